"""One-rank RCCL check of bench.py's all-gather plumbing: a zero-copy torch view of arena memory (through
__cuda_array_interface__) goes through all_gather_into_tensor and comes back intact.  (The multi-rank case needs more
than one GPU.)"""
import os
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
import torch.distributed as dist

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29611')
torch.cuda.set_device(0)
dist.init_process_group(backend='nccl', rank=0, world_size=1)
import dsc_amd as dsc
from dsc_amd import _bindings as B
from dsc_amd.context import _get_ctx
dsc.init(1 << 30, 1 << 28, device=0)
ctx = _get_ctx()
x = np.random.default_rng(0).standard_normal((64, 65536)).astype(np.float32)
X = dsc.rfft(dsc.from_numpy(x))
dsc.synchronize()
bins = 32769


class _DevView:
    def __init__(self, ptr, n_f32):
        self.__cuda_array_interface__ = {'shape': (n_f32,), 'typestr': '<f4', 'data': (ptr, False), 'version': 2}


base = X._c_ptr.contents.data
view = torch.as_tensor(_DevView(base, 64 * bins * 2), device='cuda')
recv = torch.empty((1, 64 * bins * 2), dtype=torch.float32, device='cuda')
dist.all_gather_into_tensor(recv, view)
torch.cuda.synchronize()
got = recv.cpu().numpy().reshape(64, bins, 2)
want = X.numpy()
assert np.array_equal(got[..., 0], want.real) and np.array_equal(got[..., 1], want.imag)
t = torch.tensor([1.5], dtype=torch.float64, device='cuda')
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
dist.destroy_process_group()
print('RCCL one-rank all_gather_into_tensor of an arena view: OK')
