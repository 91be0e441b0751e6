"""ctypes bindings to libdsc_mi355x.so — the C ABI in include/dsc_mi355x.h.

Same shape as the reference's python/dsc/_bindings.py:31-54 (CDLL next to the package,
explicit argtypes/restype, `_DscTensor` mirroring the 64-byte struct).  There is no CPU
fallback: a missing library or a missing GPU is an error, never a silent slow path."""
import ctypes
import os
from ctypes import POINTER, Structure, c_bool, c_char_p, c_double, c_float, c_int, c_size_t, c_uint8, c_void_p

_DSC_MAX_DIMS = 4
_DscCtx = c_void_p

# DSC_MI355X_LIB overrides the library path (A/B timing of two builds on the same box)
LIB_PATH = os.environ.get('DSC_MI355X_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libdsc_mi355x.so')
if not os.path.exists(LIB_PATH):
    raise RuntimeError(
        f'DSC MI355X backend: "{LIB_PATH}" not built. Run `make -C dsc_amd/csrc` '
        f'(or `python -c "import __graft_entry__ as g; g.build()"`). There is no CPU fallback.')

_lib = ctypes.CDLL(LIB_PATH)


class _DscTensorBuffer(Structure):
    _fields_ = [('refs', c_int)]


class _DscTensor(Structure):           # include/dsc_mi355x.h (reference: dsc/include/dsc.h:96-108)
    _fields_ = [
        ('shape', c_int * _DSC_MAX_DIMS),
        ('stride', c_int * _DSC_MAX_DIMS),
        ('buffer', POINTER(_DscTensorBuffer)),
        ('data', c_void_p),            # DEVICE pointer
        ('ne', c_int),
        ('n_dim', c_int),
        ('dtype', c_uint8),
        ('backend', c_uint8),
    ]


_DscTensor_p = POINTER(_DscTensor)


class _C32(Structure):
    _fields_ = [('real', c_float), ('imag', c_float)]


class _C64(Structure):
    _fields_ = [('real', c_double), ('imag', c_double)]


def _sig(name, restype, *argtypes):
    f = getattr(_lib, name)
    f.restype = restype
    f.argtypes = list(argtypes)
    return f


# every symbol include/dsc_mi355x.h declares (tests/test_abi.py checks the list against the header)
dsc_ctx_init = _sig('dsc_ctx_init', _DscCtx, c_size_t, c_size_t)
dsc_plan_fft = _sig('dsc_plan_fft', c_void_p, _DscCtx, c_int, c_uint8, c_uint8)
dsc_ctx_free = _sig('dsc_ctx_free', None, _DscCtx)
dsc_ctx_clear = _sig('dsc_ctx_clear', None, _DscCtx)
dsc_tensor_free = _sig('dsc_tensor_free', None, _DscCtx, _DscTensor_p)
dsc_used_mem = _sig('dsc_used_mem', c_size_t, _DscCtx)
dsc_print_mem_usage = _sig('dsc_print_mem_usage', None, _DscCtx)
dsc_new_tensor = _sig('dsc_new_tensor', _DscTensor_p, _DscCtx, c_int, POINTER(c_int), c_uint8, POINTER(_DscTensorBuffer))
dsc_view = _sig('dsc_view', _DscTensor_p, _DscCtx, _DscTensor_p)
dsc_tensor_1d = _sig('dsc_tensor_1d', _DscTensor_p, _DscCtx, c_uint8, c_int)
dsc_tensor_2d = _sig('dsc_tensor_2d', _DscTensor_p, _DscCtx, c_uint8, c_int, c_int)
dsc_tensor_3d = _sig('dsc_tensor_3d', _DscTensor_p, _DscCtx, c_uint8, c_int, c_int, c_int)
dsc_tensor_4d = _sig('dsc_tensor_4d', _DscTensor_p, _DscCtx, c_uint8, c_int, c_int, c_int, c_int)
dsc_wrap_f32 = _sig('dsc_wrap_f32', _DscTensor_p, _DscCtx, c_float)
dsc_wrap_f64 = _sig('dsc_wrap_f64', _DscTensor_p, _DscCtx, c_double)
dsc_wrap_c32 = _sig('dsc_wrap_c32', _DscTensor_p, _DscCtx, _C32)
dsc_wrap_c64 = _sig('dsc_wrap_c64', _DscTensor_p, _DscCtx, _C64)
dsc_cast = _sig('dsc_cast', _DscTensor_p, _DscCtx, _DscTensor_p, c_uint8)
dsc_mul = _sig('dsc_mul', _DscTensor_p, _DscCtx, _DscTensor_p, _DscTensor_p, _DscTensor_p)
dsc_add = _sig('dsc_add', _DscTensor_p, _DscCtx, _DscTensor_p, _DscTensor_p, _DscTensor_p)
dsc_sub = _sig('dsc_sub', _DscTensor_p, _DscCtx, _DscTensor_p, _DscTensor_p, _DscTensor_p)
dsc_div = _sig('dsc_div', _DscTensor_p, _DscCtx, _DscTensor_p, _DscTensor_p, _DscTensor_p)
dsc_abs = _sig('dsc_abs', _DscTensor_p, _DscCtx, _DscTensor_p, _DscTensor_p)
dsc_angle = _sig('dsc_angle', _DscTensor_p, _DscCtx, _DscTensor_p)
dsc_conj = _sig('dsc_conj', _DscTensor_p, _DscCtx, _DscTensor_p)
dsc_real = _sig('dsc_real', _DscTensor_p, _DscCtx, _DscTensor_p)
dsc_imag = _sig('dsc_imag', _DscTensor_p, _DscCtx, _DscTensor_p)
dsc_sum = _sig('dsc_sum', _DscTensor_p, _DscCtx, _DscTensor_p, _DscTensor_p, c_int, c_bool)
dsc_mean = _sig('dsc_mean', _DscTensor_p, _DscCtx, _DscTensor_p, _DscTensor_p, c_int, c_bool)
dsc_max = _sig('dsc_max', _DscTensor_p, _DscCtx, _DscTensor_p, _DscTensor_p, c_int, c_bool)
dsc_min = _sig('dsc_min', _DscTensor_p, _DscCtx, _DscTensor_p, _DscTensor_p, c_int, c_bool)
dsc_fft = _sig('dsc_fft', _DscTensor_p, _DscCtx, _DscTensor_p, _DscTensor_p, c_int, c_int)
dsc_ifft = _sig('dsc_ifft', _DscTensor_p, _DscCtx, _DscTensor_p, _DscTensor_p, c_int, c_int)
dsc_rfft = _sig('dsc_rfft', _DscTensor_p, _DscCtx, _DscTensor_p, _DscTensor_p, c_int, c_int)
dsc_irfft = _sig('dsc_irfft', _DscTensor_p, _DscCtx, _DscTensor_p, _DscTensor_p, c_int, c_int)


class _DscSlice(Structure):            # dsc.h:110-117, python/dsc/_bindings.py:72-73
    _fields_ = [('start', c_int), ('stop', c_int), ('step', c_int)]


DSC_VALUE_NONE = 2 ** 31 - 1           # dsc.h:78

# variadic (dsc.h:244-260): only the fixed arguments are typed, as python/dsc/_bindings.py:413-459
_dsc_tensor_get_idx = _sig('dsc_tensor_get_idx', _DscTensor_p, _DscCtx, _DscTensor_p, c_int)
_dsc_tensor_get_slice = _sig('dsc_tensor_get_slice', _DscTensor_p, _DscCtx, _DscTensor_p, c_int)
_dsc_tensor_set_idx = _sig('dsc_tensor_set_idx', None, _DscCtx, _DscTensor_p, _DscTensor_p, c_int)
_dsc_tensor_set_slice = _sig('dsc_tensor_set_slice', None, _DscCtx, _DscTensor_p, _DscTensor_p, c_int)


def dsc_tensor_get_idx(ctx, x, *indexes):
    return _dsc_tensor_get_idx(ctx, x, len(indexes), *[c_int(i) for i in indexes])


def dsc_tensor_get_slice(ctx, x, *slices):
    return _dsc_tensor_get_slice(ctx, x, len(slices), *slices)


def dsc_tensor_set_idx(ctx, xa, xb, *indexes):
    _dsc_tensor_set_idx(ctx, xa, xb, len(indexes), *[c_int(i) for i in indexes])


def dsc_tensor_set_slice(ctx, xa, xb, *slices):
    _dsc_tensor_set_slice(ctx, xa, xb, len(slices), *slices)


_dsc_transpose = _sig('dsc_transpose', _DscTensor_p, _DscCtx, _DscTensor_p, c_int)


def dsc_transpose(ctx, x, *axes):
    return _dsc_transpose(ctx, x, len(axes), *[c_int(a) for a in axes])


dsc_fftfreq = _sig('dsc_fftfreq', _DscTensor_p, _DscCtx, c_int, c_double, c_uint8)
dsc_rfftfreq = _sig('dsc_rfftfreq', _DscTensor_p, _DscCtx, c_int, c_double, c_uint8)
dsc_traces_record = _sig('dsc_traces_record', None, _DscCtx, c_bool)
dsc_dump_traces = _sig('dsc_dump_traces', None, _DscCtx, c_char_p)
dsc_clear_traces = _sig('dsc_clear_traces', None, _DscCtx)
dsc_set_device = _sig('dsc_set_device', c_int, c_int)
dsc_copy_from_host = _sig('dsc_copy_from_host', None, _DscCtx, _DscTensor_p, c_void_p, c_size_t)
dsc_copy_to_host = _sig('dsc_copy_to_host', None, _DscCtx, _DscTensor_p, c_void_p, c_size_t)
dsc_synchronize = _sig('dsc_synchronize', None, _DscCtx)
dsc_stream = _sig('dsc_stream', c_void_p, _DscCtx)
dsc_timer_start = _sig('dsc_timer_start', None, _DscCtx)
dsc_timer_stop = _sig('dsc_timer_stop', c_float, _DscCtx)
dsc_filter_fft = _sig('dsc_filter_fft', _DscTensor_p, _DscCtx, _DscTensor_p, _DscTensor_p, _DscTensor_p)
dsc_last_fft_path = _sig('dsc_last_fft_path', c_char_p, _DscCtx)


class _DscIpcHandle(Structure):        # include/dsc_mi355x.h section C
    _fields_ = [('bytes', c_uint8 * 64)]


dsc_device_alloc = _sig('dsc_device_alloc', c_void_p, _DscCtx, c_size_t)
dsc_device_free = _sig('dsc_device_free', None, _DscCtx, c_void_p)
dsc_tensor_from_device_ptr = _sig('dsc_tensor_from_device_ptr', _DscTensor_p, _DscCtx, c_void_p, c_size_t, c_int, POINTER(c_int), c_uint8)
dsc_ipc_export = _sig('dsc_ipc_export', c_int, _DscCtx, c_void_p, POINTER(_DscIpcHandle))
dsc_ipc_open = _sig('dsc_ipc_open', c_void_p, _DscCtx, POINTER(_DscIpcHandle))
dsc_ipc_close = _sig('dsc_ipc_close', c_int, _DscCtx, c_void_p)
dsc_peer_lanes = _sig('dsc_peer_lanes', c_int)
dsc_peer_push = _sig('dsc_peer_push', c_int, _DscCtx, c_void_p, c_void_p, c_size_t, c_int)
dsc_peer_wait = _sig('dsc_peer_wait', c_int, _DscCtx)


class _DscCommId(Structure):           # ncclUniqueId as bytes
    _fields_ = [('bytes', c_uint8 * 128)]


_DscComm = c_void_p
dsc_comm_unique_id = _sig('dsc_comm_unique_id', c_int, POINTER(_DscCommId))
dsc_comm_init_rank = _sig('dsc_comm_init_rank', _DscComm, _DscCtx, POINTER(_DscCommId), c_int, c_int)
dsc_comm_n_ranks = _sig('dsc_comm_n_ranks', c_int, _DscComm)
dsc_comm_rank = _sig('dsc_comm_rank', c_int, _DscComm)
dsc_comm_free = _sig('dsc_comm_free', None, _DscComm)
dsc_shard_allgather = _sig('dsc_shard_allgather', c_int, _DscCtx, _DscComm, c_void_p, c_size_t, c_size_t)
dsc_shard_exchange_rows = _sig('dsc_shard_exchange_rows', c_int, _DscCtx, _DscComm, c_void_p, c_size_t, c_size_t, c_size_t, c_size_t)

EXPORTS = [n for n in dir() if n.startswith('dsc_') and n != 'dsc_api']
