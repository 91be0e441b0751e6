"""One-rank check of dsc_amd/shard.py's 'allgather_c' / 'p2p_c': the library's own RCCL communicator and collectives
(include/dsc_mi355x.h section C: dsc_comm_init_rank, dsc_shard_allgather, dsc_shard_exchange_rows) driven through ShardGather
exactly as bench.py drives them with more ranks.  The transform writes its slot of the destination in place; afterwards the
destination must equal the transform's own output bit for bit and verify() must pass."""
import ctypes
import sys
sys.path.insert(0, '.')
import numpy as np
import torch

torch.cuda.set_device(0)
import dsc_amd as dsc
from dsc_amd import _bindings as B, shard
from dsc_amd.context import _get_ctx


class Solo:                                   # the slice of torch.distributed a one-rank ShardGather touches
    @staticmethod
    def get_world_size():
        return 1

    @staticmethod
    def get_rank():
        return 0

    @staticmethod
    def barrier():
        pass


dsc.init(1 << 30, 1 << 28, device=0)
ctx = _get_ctx()
rows, n = 64, 4096
bins = n // 2 + 1
x = np.random.default_rng(8).standard_normal((rows, n)).astype(np.float32)
want = dsc.rfft(dsc.from_numpy(x)).numpy()
for method in ('allgather_c', 'p2p_c'):
    dest = shard.DeviceDest(ctx, 1, 0, rows, 2 * bins)
    g = shard.ShardGather(Solo, dest.tensor, 24, method=method, ctx=ctx, dest_ptr=dest.ptr)
    tx = dsc.from_numpy(x)
    for i, (r0, m) in enumerate(g.chunks):
        shp = (ctypes.c_int * 2)(m, n)
        xin = B.dsc_tensor_from_device_ptr(ctx, tx._c_ptr.contents.data + r0 * n * 4, m * n * 4, 2, shp, int(dsc.Dtype.F32))
        B.dsc_rfft(ctx, xin, dest.own_slot_tensor(r0, m, dsc.Dtype.C32, bins), -1, -1)
        B.dsc_tensor_free(ctx, xin)
        g.push(i)
    g.finish()
    got = dest.tensor.cpu().numpy().reshape(rows, bins, 2)
    assert np.array_equal(got[..., 0], want.real) and np.array_equal(got[..., 1], want.imag), method
    assert g.verify()['verified'] is True, method
    g.close()
    dest.free()
print("shard.py 'allgather_c' / 'p2p_c' with one rank: OK")
