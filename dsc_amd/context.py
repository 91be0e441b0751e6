"""`dsc.init()` — the context singleton (mirror of python/dsc/context.py:13-51).

The arenas are HBM: sizes default to a slice of the device's memory rather than of host
RAM as in the reference (context.py:18-19)."""
import os

from . import _bindings as B

_ctx_instance = None


def _get_ctx():
    global _ctx_instance
    if _ctx_instance is None:
        mem = 2 << 30
        print(f'DSC has not been explicitly initialized. Using {mem >> 20}MB of HBM for both the main and scratch '
              f'memory. If you require more memory please call dsc.init() once before executing your code.')
        _ctx_instance = _DscContext(mem, mem)
    return _ctx_instance._ctx


def init(main_mem: int, scratch_mem: int, device: int = None):
    """Create the context: `main_mem` + `scratch_mem` bytes of HBM on `device`
    (default: $LOCAL_RANK, else 0 — one process per GPU)."""
    global _ctx_instance
    if _ctx_instance is not None:
        raise RuntimeWarning('Context already initialized')
    if device is None:
        device = int(os.environ.get('LOCAL_RANK', '0'))
    B.dsc_set_device(device)
    _ctx_instance = _DscContext(main_mem, scratch_mem)


def _current():
    """The live context object or None — never creates one (Tensor.__del__ uses this)."""
    return _ctx_instance


def clear():
    if _ctx_instance is not None:
        _ctx_instance.clear()


def shutdown():
    """Free the context (the reference relies on interpreter teardown)."""
    global _ctx_instance
    _ctx_instance = None


def synchronize():
    B.dsc_synchronize(_get_ctx())


def used_mem() -> int:
    return B.dsc_used_mem(_get_ctx())


def last_fft_path() -> str:
    return B.dsc_last_fft_path(_get_ctx()).decode()


class _DscContext:
    def __init__(self, main_mem: int, scratch_mem: int):
        self._ctx = B.dsc_ctx_init(main_mem, scratch_mem)
        self.epoch = 0          # bumped by clear(): handles created before a clear are dead and must not be freed again

    def __del__(self):
        if B is not None and self._ctx:
            B.dsc_ctx_free(self._ctx)
            self._ctx = None

    def clear(self):
        self.epoch += 1
        B.dsc_ctx_clear(self._ctx)
