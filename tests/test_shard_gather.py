"""Reassembly of batch-sharded outputs (dsc_amd/shard.py) on CPU: real gloo ranks, real deterministic
per-rank arrays, every method that runs without a GPU.  Asserts gathered == concatenated shards (order, offsets,
ragged last chunk), that verify() says so, and that verify() notices a corrupted slot."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def shard_of(rank, rows, row_elems):
    """Deterministic per-rank shard: every element encodes (rank, row, column)."""
    j = np.arange(rows, dtype=np.float32)[:, None]
    c = np.arange(row_elems, dtype=np.float32)[None, :]
    return (1000.0 * (rank + 1) + j + c / 1024.0).astype(np.float32)


def _worker(rank, world, port, method, rows, row_elems, chunk_rows, corrupt, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch
    import torch.distributed as dist
    from dsc_amd import shard
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        dest = torch.full((world, rows, row_elems), -1.0, dtype=torch.float32)
        g = shard.ShardGather(dist, dest, chunk_rows, method=method)
        mine = torch.from_numpy(shard_of(rank, rows, row_elems))
        for i, (r0, n) in enumerate(g.chunks):           # "transform" chunk i into the own slot, then push it
            dest[rank, r0:r0 + n] = mine[r0:r0 + n]
            g.push(i)
        g.finish()
        want = np.stack([shard_of(r, rows, row_elems) for r in range(world)])
        equal = bool(np.array_equal(dest.numpy(), want))
        v = g.verify()
        v_bad = None
        if corrupt:
            if rank == world - 1:
                dest[0, rows // 2, 3] += 1.0               # one element of another rank's slot, on one rank only
            v_bad = g.verify()
        g.close()
        q.put((rank, equal, v, v_bad, len(g.chunks)))
    finally:
        dist.destroy_process_group()


def _run(world, method, rows, row_elems, chunk_rows, corrupt=False):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, method, rows, row_elems, chunk_rows, corrupt, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return sorted(out)


@pytest.mark.parametrize('method', ['p2p', 'allgather'])
def test_two_ranks_gathered_equals_concatenated(method):
    res = _run(2, method, rows=50, row_elems=66, chunk_rows=16)       # 16+16+16+2: ragged last chunk
    for rank, equal, v, _, n_chunks in res:
        assert n_chunks == 4
        assert equal, f'rank {rank}: gathered != concatenated shards ({method})'
        assert v['verified'] is True and v['bad_slots_on_this_rank'] == []


def test_three_ranks_p2p_and_corruption_is_noticed():
    res = _run(3, 'p2p', rows=9, row_elems=10, chunk_rows=4, corrupt=True)
    for rank, equal, v, v_bad, _ in res:
        assert equal and v['verified'] is True
        assert v_bad['verified'] is False                 # the MIN all-reduce tells every rank
    assert [r for r, *_ in res if _bad_slots(res, r)] == [2]


def _bad_slots(res, rank):
    return [x for x in res if x[0] == rank][0][3]['bad_slots_on_this_rank']


def _uneven_worker(rank, world, port, method, total, row_elems, chunk_rows, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch
    import torch.distributed as dist
    from dsc_amd import shard
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        whole = shard_of(0, total, row_elems)                       # the global array; rank r transforms its block of it
        start, count = shard.block_partition(total, world, rank)
        rows = shard.slot_rows(total, world)
        dest = torch.full((world, rows, row_elems), -1.0, dtype=torch.float32)
        g = shard.ShardGather(dist, dest, chunk_rows, method=method, valid_rows=count)
        for i, (r0, n) in enumerate(g.chunks):
            m = max(0, min(n, count - r0))
            dest[rank, r0:r0 + m] = torch.from_numpy(whole[start + r0:start + r0 + m])
            g.push(i)
        g.finish()
        equal = bool(np.array_equal(g.gathered_rows().numpy(), whole))
        v = g.verify()
        # a rank that sizes its slot from its OWN count (the mistake the agreement check exists for) must raise everywhere
        raised = False
        try:
            shard.ShardGather(dist, torch.zeros((world, count if rank else rows, row_elems)), chunk_rows, method=method)
        except ValueError:
            raised = True
        q.put((rank, equal, v, g.valid, raised))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('method', ['p2p', 'allgather'])
def test_uneven_shards_keep_their_rows(method):
    """11 rows over 3 ranks = 4 + 4 + 3: slots of ceil(11 / 3) rows, the short shard says how many of its rows count;
    gathered valid rows == the global array on every rank; mismatched slot shapes raise on every rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_uneven_worker, args=(r, 3, port, method, 11, 6, 3, q)) for r in range(3)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, equal, v, valid, raised in out:
        assert valid == [4, 4, 3]
        assert equal, f'rank {rank}: gathered valid rows != global array'
        assert v['verified'] is True
        assert raised, f'rank {rank}: mismatched slot shapes went unnoticed'


def test_block_partition_covers_every_row_once():
    sys.path.insert(0, ROOT)
    from dsc_amd import shard
    for total, world in ((65536, 8), (8192, 1), (10, 3), (7, 8)):
        spans = [shard.block_partition(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == total
        for (s0, c0), (s1, _) in zip(spans, spans[1:]):
            assert s0 + c0 == s1
    assert shard.block_partition(65536, 8, 3) == (3 * 8192, 8192)     # config 4: 8 shards of config 2
    assert shard.slot_rows(11, 3) == 4 and shard.slot_rows(65536, 8) == 8192
    assert shard.chunk_bounds(50, 16) == [(0, 16), (16, 16), (32, 16), (48, 2)]
