"""dsc_amd/shard.py — batch sharding of independent 1-D transforms across the GPUs of one node, and the
reassembly of the output shards (BASELINE config 4 = 8 shards of config 2).

The reference has no counterpart: it has one backend and no communication layer (dsc/include/dsc_backend.h:11-13).
Every transform along the last axis is independent (dsc/src/dsc.cpp:2124-2143 loops over lines), so rank r — one
process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI, "gloo" on CPU) — transforms rows
[r*B/P, (r+1)*B/P) with NO collective on the data path.  What this module adds is the step after it: putting the
P shards next to each other on every rank.

Destination layout: ONE persistent buffer per rank, `dest[P][rows][row_elems]`, rank-major — i.e. exactly the
concatenation of the shards in rank order (global row g = r*rows + j lives at dest[r][j]).  It is a raw buffer, not
a dsc_tensor: config 4's 65536 x 32769 c32 exceeds `int ne` (dsc/include/dsc.h:104).  A rank's own slot dest[rank]
is where its transform writes (`own_slot_tensor`), so the local shard is never copied.

Three exchange methods, all filling the same layout:

  'allgather'  ONE all_gather_into_tensor of the whole shard, in place (input = dest[rank]) — the plain RCCL
               all-gather north_star names.  RCCL's ring moves (P-1) steps over one link each: ~(P-1)*S/153 GB/s.
  'p2p'        per chunk of rows, one group of P-1 sends + P-1 receives (batch_isend_irecv = grouped
               ncclSend/ncclRecv): every GPU talks to all peers at once, one xGMI link per peer (~S/153 GB/s),
               and chunk i travels while chunk i+1 is being transformed.
  'allgather_c', 'p2p_c'   the same two RCCL patterns through the library's own C entry points (`dsc_comm_init_rank`,
               `dsc_shard_allgather`, `dsc_shard_exchange_rows`: include/dsc_mi355x.h section C) — what a C++ host of the
               reference's API calls; torch.distributed only carries the 128-byte unique id to the other ranks.
  'ipc'        per chunk, P-1 direct copies into the peers' destinations, which are mapped into this process
               through HIP IPC (dsc_ipc_export / dsc_ipc_open) — `dsc_peer_push`, one copy stream per peer,
               ordered after the transform on the context's stream by an event.  No RCCL involved: the
               hand-rolled comparator SURVEY 8e asks for.  Device buffers only.

Ordering on a GPU: pushes are issued with the context's stream as the current stream, so RCCL's internal stream (or
the copy lane) waits for the transform that produced the chunk, and nothing waits for the exchange until `finish()`.

`verify()` proves the reassembly: every rank checks dest[r] against a position-weighted checksum and sample rows that
rank r computed from its own shard — gathered == concatenated shards, bit for bit.
"""
import ctypes

METHODS = ('allgather', 'p2p', 'ipc', 'allgather_c', 'p2p_c')


def block_partition(total_rows: int, world: int, rank: int):
    """Leading-axis block partition: rank r owns rows [start, start + count).  The first
    `total_rows % world` ranks take one extra row, so any batch size shards.  When the counts differ, the destination's
    slots are `slot_rows(total_rows, world)` rows each and ShardGather is told this rank's `valid_rows`."""
    base, extra = divmod(total_rows, world)
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


def slot_rows(total_rows: int, world: int) -> int:
    """Rows of one slot of the [world][slot][row_elems] destination: the largest shard (ceil)."""
    return -(-total_rows // world)


def chunk_bounds(rows: int, chunk_rows: int):
    """[(first_row, n_rows)] of the chunks a shard travels in; the last one may be ragged."""
    return [(r0, min(chunk_rows, rows - r0)) for r0 in range(0, rows, chunk_rows)]


class _DevView:
    """Zero-copy torch view of raw device memory (`torch.as_tensor(_DevView(...), device='cuda')`)."""

    def __init__(self, ptr, n_f32):
        self.__cuda_array_interface__ = {'shape': (n_f32,), 'typestr': '<f4', 'data': (ptr, False), 'version': 2}


class DeviceDest:
    """The [world][rows][row_elems] f32 destination in HBM: dsc_device_alloc memory (exportable over HIP IPC),
    a torch view of it for the collectives, and dsc_tensor headers over this rank's own slot."""

    def __init__(self, ctx, world, rank, rows, row_elems):
        import torch
        from . import _bindings as B
        self._B, self.ctx = B, ctx
        self.world, self.rank, self.rows, self.row_elems = world, rank, rows, row_elems
        self.nbytes = world * rows * row_elems * 4
        self.ptr = B.dsc_device_alloc(ctx, self.nbytes)
        if not self.ptr:
            raise MemoryError(f'dsc_device_alloc({self.nbytes}) failed')
        self.tensor = torch.as_tensor(_DevView(self.ptr, world * rows * row_elems), device='cuda').view(world, rows, row_elems)
        self._headers = []

    def slot_ptr(self, r, row=0):
        return self.ptr + ((r * self.rows + row) * self.row_elems) * 4

    def own_slot_tensor(self, row0, n_rows, dtype, cols):
        """dsc_tensor [n_rows, cols] of `dtype` over rows [row0, row0 + n_rows) of this rank's slot: the transform
        writes its output here, in place."""
        B = self._B
        es = {0: 4, 1: 8, 2: 8, 3: 16}[int(dtype)]
        assert cols * es == self.row_elems * 4, 'row size of the tensor must equal the destination row'
        shape = (ctypes.c_int * 2)(n_rows, cols)
        t = B.dsc_tensor_from_device_ptr(self.ctx, self.slot_ptr(self.rank, row0), n_rows * cols * es, 2, shape, int(dtype))
        self._headers.append(t)
        return t

    def free(self):
        B = self._B
        for t in self._headers:
            B.dsc_tensor_free(self.ctx, t)
        self._headers = []
        if self.ptr:
            self.tensor = None
            B.dsc_device_free(self.ctx, self.ptr)
            self.ptr = None


class ShardGather:
    """Chunk-wise exchange of this rank's shard into every rank's `dest` (see the module docstring).

    dest        torch tensor [world, rows, row_elems] on any device (CPU for gloo / host staging, HBM for RCCL);
                this rank's shard is (being) written to dest[rank]
    chunk_rows  rows per exchange step ('p2p' and 'ipc'); 'allgather' moves the whole shard in finish()
    ctx         dsc context (GPU only): its stream orders the pushes after the transforms; required for 'ipc'
    dest_ptr    raw device pointer of `dest` ('ipc' only)
    valid_rows  rows of this rank's shard when it is shorter than the slot (uneven block_partition); default: the whole slot.
                Whole slots travel (equal-size collectives); `valid` holds every rank's count, `verify()` and
                `gathered_rows()` look at the valid rows only.

    Every rank must pass the same slot shape and chunking: checked here with one all_gather_object, so that a mismatch
    raises on every rank instead of mis-laying rows or hanging a collective.
    """

    def __init__(self, dist, dest, chunk_rows, method='p2p', ctx=None, dest_ptr=None, valid_rows=None):
        if method not in METHODS:
            raise ValueError(f'method must be one of {METHODS}')
        self.dist, self.dest, self.method, self.ctx = dist, dest, method, ctx
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        assert dest.dim() == 3 and dest.shape[0] == self.world and dest.is_contiguous()
        self.rows, self.row_elems = int(dest.shape[1]), int(dest.shape[2])
        mine = self.rows if valid_rows is None else int(valid_rows)
        if not 0 <= mine <= self.rows:
            raise ValueError(f'valid_rows = {mine} outside the slot of {self.rows} rows')
        said = [None] * self.world
        if self.world > 1:
            dist.all_gather_object(said, (self.rows, self.row_elems, int(chunk_rows), mine))
        else:
            said[0] = (self.rows, self.row_elems, int(chunk_rows), mine)
        if any(x[:3] != said[0][:3] for x in said):
            raise ValueError('ShardGather: (slot rows, row elements, chunk rows) differ between ranks: ' + repr([x[:3] for x in said]) +
                             ' — size the slots with shard.slot_rows() and pass valid_rows for the shorter shards')
        self.valid = [x[3] for x in said]
        self.chunks = chunk_bounds(self.rows, chunk_rows)
        self.on_gpu = dest.is_cuda
        self._works = []
        self._pushed = 0
        self._stream = None
        if self.on_gpu and ctx is not None:
            import torch
            from . import _bindings as B
            self._stream = torch.cuda.ExternalStream(B.dsc_stream(ctx))
        self._mapped = None
        self._failed = None
        self._comm = None
        if method in ('allgather_c', 'p2p_c'):
            if not (self.on_gpu and ctx is not None and dest_ptr):
                raise ValueError(f"'{method}' needs a device destination, its raw pointer and the dsc context")
            self._own_ptr = dest_ptr
            self._open_comm()
        if method == 'ipc':
            if not (self.on_gpu and ctx is not None and dest_ptr):
                raise ValueError("'ipc' needs a device destination, its raw pointer and the dsc context")
            self._open_peers(dest_ptr)

    # ---- HIP IPC: map every peer's destination into this process
    def _open_peers(self, dest_ptr):
        """Every step that can fail on ONE rank only (export, opening a peer's handle) is followed by an agreement among all
        ranks, so that a failure makes every rank raise together instead of leaving the others waiting in a collective."""
        from . import _bindings as B
        h = B._DscIpcHandle()
        exported = B.dsc_ipc_export(self.ctx, dest_ptr, ctypes.byref(h)) == 0
        handles = [None] * self.world
        self.dist.all_gather_object(handles, bytes(h.bytes) if exported else None)
        failed = [p for p in range(self.world) if handles[p] is None]
        if failed:
            raise RuntimeError(f'dsc_ipc_export failed on rank(s) {failed}')
        mapped = [None] * self.world
        bad_peer = None
        for p in range(self.world):
            if p == self.rank:
                mapped[p] = dest_ptr
                continue
            hp = B._DscIpcHandle()
            ctypes.memmove(hp.bytes, handles[p], 64)
            m = B.dsc_ipc_open(self.ctx, ctypes.byref(hp))
            if not m:
                bad_peer = p
                break
            mapped[p] = m
        verdicts = [None] * self.world
        self.dist.all_gather_object(verdicts, bad_peer)
        if any(v is not None for v in verdicts):
            for p, m in enumerate(mapped):
                if p != self.rank and m:
                    B.dsc_ipc_close(self.ctx, m)
            raise RuntimeError('dsc_ipc_open failed: ' + ', '.join(f'rank {r} could not map rank {v}' for r, v in enumerate(verdicts) if v is not None))
        self._mapped = mapped
        self._own_ptr = dest_ptr

    # ---- the library's own RCCL communicator (include/dsc_mi355x.h section C)
    def _open_comm(self):
        """Rank 0 makes the unique id, torch.distributed's object broadcast ships it, every rank joins; a rank that fails
        says so to all before anybody raises."""
        from . import _bindings as B
        box = [None]
        if self.rank == 0:
            cid = B._DscCommId()
            box[0] = bytes(cid.bytes) if B.dsc_comm_unique_id(ctypes.byref(cid)) == 0 else None
        if self.world > 1:
            self.dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            raise RuntimeError('dsc_comm_unique_id failed on rank 0 (RCCL not loadable?)')
        cid = B._DscCommId()
        ctypes.memmove(cid.bytes, box[0], 128)
        comm = B.dsc_comm_init_rank(self.ctx, ctypes.byref(cid), self.world, self.rank)
        said = [None] * self.world
        if self.world > 1:
            self.dist.all_gather_object(said, bool(comm))
        else:
            said[0] = bool(comm)
        if not all(said):
            if comm:
                B.dsc_comm_free(comm)
            raise RuntimeError(f'dsc_comm_init_rank failed on rank(s) {[r for r, ok in enumerate(said) if not ok]}')
        self._comm = comm

    def _under_ctx_stream(self):
        import contextlib
        import torch
        return torch.cuda.stream(self._stream) if self._stream is not None else contextlib.nullcontext()

    def push(self, i):
        """Start the exchange of chunk i of this rank's shard (asynchronous).  On a GPU it is ordered after
        everything enqueued so far on the context's stream."""
        r0, n = self.chunks[i]
        self._pushed += 1
        if self.method in ('allgather', 'allgather_c') or self.world == 1:
            return
        if self.method == 'p2p_c':
            from . import _bindings as B
            if B.dsc_shard_exchange_rows(self.ctx, self._comm, self._own_ptr, self.rows, self.row_elems * 4, r0, n) != 0:
                self._failed = self._failed or 'dsc_shard_exchange_rows failed'
            return
        if self.method == 'ipc':
            from . import _bindings as B
            lanes = B.dsc_peer_lanes()
            off = ((self.rank * self.rows + r0) * self.row_elems) * 4
            nbytes = n * self.row_elems * 4
            for k in range(1, self.world):
                p = (self.rank + k) % self.world          # staggered: at step k every rank targets a different peer
                if B.dsc_peer_push(self.ctx, self._mapped[p] + off, self._own_ptr + off, nbytes, (k - 1) % lanes) != 0:
                    self._failed = self._failed or f'dsc_peer_push to rank {p} failed'    # reported by finish(), on every rank
            return
        # 'p2p': one group of sends and receives per chunk
        dist = self.dist
        mine = self.dest[self.rank, r0:r0 + n]
        ops = []
        for k in range(1, self.world):
            to = (self.rank + k) % self.world
            frm = (self.rank - k) % self.world
            ops.append(dist.P2POp(dist.isend, mine, to))
            ops.append(dist.P2POp(dist.irecv, self.dest[frm, r0:r0 + n], frm))
        with self._under_ctx_stream():
            self._works.extend(dist.batch_isend_irecv(ops))

    def finish(self):
        """Complete every exchange started by push() (for 'allgather': run it), then a barrier: on return every
        rank's dest holds all shards."""
        assert self._pushed == len(self.chunks), 'finish() before every chunk was pushed'
        dist = self.dist
        if self.method == 'allgather_c':                      # also with one rank: the C entry point and RCCL are exercised for real
            from . import _bindings as B
            if B.dsc_shard_allgather(self.ctx, self._comm, self._own_ptr, self.rows, self.row_elems * 4) != 0:
                self._failed = self._failed or 'dsc_shard_allgather failed'
        if self.method in ('allgather_c', 'p2p_c'):
            from . import _bindings as B
            B.dsc_synchronize(self.ctx)                       # the collectives were enqueued on the context's stream
        if self.world > 1:
            if self.method == 'allgather':
                with self._under_ctx_stream():
                    w = dist.all_gather_into_tensor(self.dest.view(-1), self.dest[self.rank].reshape(-1), async_op=True)
                self._works.append(w)
            for w in self._works:
                w.wait()
            self._works = []
            if self.method == 'ipc':
                from . import _bindings as B
                if B.dsc_peer_wait(self.ctx) != 0:
                    self._failed = self._failed or 'dsc_peer_wait failed'
        if self.on_gpu:
            import torch
            torch.cuda.synchronize()
            if self.ctx is not None:                          # also the library's own check of its deferred device-side errors
                from . import _bindings as B
                B.dsc_synchronize(self.ctx)
        self._pushed = 0
        if self.world > 1:
            if self.method in ('ipc', 'allgather_c', 'p2p_c'):   # the barrier doubles as the agreement on a one-sided failure
                said = [None] * self.world
                dist.all_gather_object(said, self._failed)
                self._failed = None
                if any(said):
                    raise RuntimeError('; '.join(f'rank {r}: {m}' for r, m in enumerate(said) if m))
            else:
                dist.barrier()
        elif self._failed:
            msg, self._failed = self._failed, None
            raise RuntimeError(msg)

    def close(self):
        if self._comm is not None:
            from . import _bindings as B
            B.dsc_comm_free(self._comm)
            self._comm = None
        if self._mapped is not None:
            from . import _bindings as B
            self.dist.barrier()                               # nobody is still writing into a mapping
            for p, m in enumerate(self._mapped):
                if p != self.rank and m:
                    B.dsc_ipc_close(self.ctx, m)
            self._mapped = None

    # ---- proof of reassembly
    def _digest(self, shard, valid):
        """(position-weighted checksum of the shard's valid rows, sample rows) — integer arithmetic on the raw bits."""
        import torch
        bits = shard.view(torch.int32)[:valid]
        row_sums = torch.sum(bits, dim=1, dtype=torch.int64)
        weights = torch.arange(1, valid + 1, dtype=torch.int64, device=shard.device)
        cks = int(torch.sum(row_sums * weights).item())       # int64 wrap-around is deterministic
        idx = sorted(i for i in ({r0 for r0, _ in self.chunks} | {r0 + n - 1 for r0, n in self.chunks}) if i < valid)
        samples = bits[idx].cpu()
        return cks, idx, samples

    def gathered_rows(self):
        """The concatenation of the shards' valid rows, rank order — a copy; equals dest.view(-1, row_elems) when every
        shard fills its slot."""
        import torch
        return torch.cat([self.dest[r, :self.valid[r]] for r in range(self.world)], dim=0)

    def verify(self):
        """Every rank checks every slot of its dest against what the owning rank says its shard is.
        Returns {'verified': bool, ...}; never raises on a mismatch (the caller reports it)."""
        import torch
        mine = self._digest(self.dest[self.rank], self.valid[self.rank])
        every = [None] * self.world
        if self.world > 1:
            self.dist.all_gather_object(every, mine)
        else:
            every[0] = mine
        bad = []
        for r in range(self.world):
            cks, idx, samples = self._digest(self.dest[r], self.valid[r])
            want_cks, want_idx, want_samples = every[r]
            if cks != want_cks or idx != want_idx or not torch.equal(samples, want_samples):
                bad.append(r)
        on_rccl = self.world > 1 and self.dist.get_backend() == 'nccl'
        ok = torch.tensor([0 if bad else 1], dtype=torch.int32, device='cuda' if on_rccl else 'cpu')
        if self.world > 1:
            self.dist.all_reduce(ok, op=self.dist.ReduceOp.MIN)
        return {'verified': bool(ok.item() == 1), 'bad_slots_on_this_rank': bad, 'rows_sampled_per_slot': len(mine[1]),
                'checksum': 'sum_j (j+1) * sum(int32 bits of row j), int64'}
