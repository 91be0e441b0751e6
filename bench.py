#!/usr/bin/env python3
"""bench.py — batched 1-D rfft throughput on MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one pass of the hot path over one batch already resident in HBM:
dsc_rfft of a [8192, 65536] f32 tensor into a [8192, 32769] c32 tensor through the C ABI
(BASELINE.json configs[1]).  With N > 1 (launched by torch.distributed.run, one process per
GPU) every rank transforms its own 8192-row shard — independent rows, no collective on the
data path — and `value` is the rows of all ranks over the slowest rank's time (weak scaling:
config 4 is exactly 8 shards of config 2).  The RCCL all-gather that reassembles the shards
is timed as its own phase, outside the timed region, and reported beside the metric
(`allgather`), because it is xGMI-bound at >= 14 ms against ~1 ms of transform.

Prints ONE JSON line on rank 0.  Extra keys beyond the driver's contract:
  roofline      dominant kernel vs the 8 TB/s HBM peak (algorithmic bytes / HIP-event time); `traffic` is the PMC figure
                recorded under profiles/ for the same kernel and size (`traffic_source` says where / when; null if none)
  other_kernels irfft, fused filter and config 5, each a full roofline object (same timer, never part of `value`)
  allgather     the shard-reassembly phase (dsc_amd/shard.py): every method timed and verified (N > 1 only)
  cpu_baseline  the reference's own CPU code (oracle/_ref, kind "reference") or our C
                restatement (kind "port") timed on this host, rank 0, N == 1 only
  parity        SURVEY 8d's on-device check (untimed): first / last 4 rows + 64 seeded random rows against the CPU oracle, and
                Parseval's identity on ALL rows, evaluated on the device with the product's own dsc_mul / dsc_conj / dsc_sum
  min_ms, median_ms   per-step HIP-event times of the K timed steps (the reference reports the minimum, utils.py:11-12)
  config4_on_one_gpu  N == 1 only: the 65536 rows of BASELINE configs[3] as 8 sequential 8192-row launches on this GPU — the
                like-for-like denominator of the 8-GPU line
"""
import argparse
import contextlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_FFT = 65536
BATCH = 8192
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
BYTES_PER_ROW = N_FFT * 4 + (N_FFT // 2 + 1) * 8      # 524,296 B: 4 B/sample in + 8 B/bin out


@contextlib.contextmanager
def c_stdout_to_stderr():
    """The C libraries log to stdout (the reference's convention, dsc.h:20); keep stdout for
    the one JSON line."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def synthetic_rows(rows, rank):
    import numpy as np
    rng = np.random.default_rng(1234 + rank)
    return rng.standard_normal((rows, N_FFT), dtype=np.float32)


def cpu_baseline(rows=1024, reps=5, warm=2):
    """Reference convention: 2 warm-ups, min of 5 (benchmarks/python/utils.py:11-12), 1 thread
    (the reference is single threaded), on a bounded sample of the same workload."""
    import numpy as np
    x = synthetic_rows(rows, 0)
    kind = 'port'
    with c_stdout_to_stderr():
        try:
            from oracle import ref
            if not ref.available():
                raise RuntimeError('no _ref')
            R = ref.Ref.get(main_mem=rows * BYTES_PER_ROW + (64 << 20), scratch_mem=16 << 20)
            tx = R.put(x)
            tout = R.L.dsc_tensor_2d(R.ctx, ref.C32, rows, N_FFT // 2 + 1)
            run = lambda: R.rfft_raw(tx, tout)      # noqa: E731
            kind = 'reference'
        except Exception:
            from oracle import port
            run = lambda: port.rfft(x)              # noqa: E731
        for _ in range(warm):
            run()
        best = float('inf')
        for _ in range(reps):
            t0 = time.perf_counter()
            run()
            best = min(best, time.perf_counter() - t0)
    return {
        'value': round(rows * N_FFT / best / 1e9, 5),
        'unit': 'GSamples/s',
        'cores': 1,
        'kind': kind,
        'sample': f'{rows} rows of the same [8192, 65536] f32 workload, min of {reps} after {warm} warm-ups, '
                  f'{best * 1e3:.1f} ms per pass, host {os.cpu_count()} logical CPUs',
    }


def device_parity(dsc, B, ctx, x, out, x_host, rows, rank):
    """SURVEY 8d, untimed.  (1) first / last 4 rows + 64 seeded random rows of the spectrum on the device against the CPU oracle
    (rel-L2 per row, tolerance 1e-5).  (2) Parseval on EVERY row with the product's own operators:
    sum_n x[n]^2 == (2 sum_k |X[k]|^2 - |X[0]|^2 - |X[N/2]|^2) / N, left side dsc_sum(dsc_mul(x, x)), right side
    dsc_sum(dsc_mul(X, dsc_conj(X))) — 1024 rows at a time through views, so that the temporaries stay under 1 GiB."""
    import ctypes
    import numpy as np
    from oracle import port                                   # checker only
    bins = N_FFT // 2 + 1
    rng = np.random.default_rng(4321 + rank)
    pick = sorted(set(list(range(min(4, rows))) + list(range(max(0, rows - 4), rows)) + [int(r) for r in rng.integers(0, rows, size=min(64, rows))]))
    x_ptr, o_ptr = x._c_ptr.contents.data, out._c_ptr.contents.data
    got = np.empty((len(pick), bins), np.complex64)
    shp1 = (ctypes.c_int * 2)(1, bins)
    for i, r in enumerate(pick):
        v = B.dsc_tensor_from_device_ptr(ctx, o_ptr + r * bins * 8, bins * 8, 2, shp1, int(dsc.Dtype.C32))
        B.dsc_copy_to_host(ctx, v, got[i].ctypes.data, bins * 8)
        B.dsc_tensor_free(ctx, v)
    want = port.rfft(x_host[pick])
    per_row = np.linalg.norm(got - want, axis=1) / np.linalg.norm(want, axis=1)
    # Parseval, all rows, on the device
    worst = 0.0
    step = min(1024, rows)
    for r0 in range(0, rows, step):
        n = min(step, rows - r0)
        sx = (ctypes.c_int * 2)(n, N_FFT)
        so = (ctypes.c_int * 2)(n, bins)
        xv = B.dsc_tensor_from_device_ptr(ctx, x_ptr + r0 * N_FFT * 4, n * N_FFT * 4, 2, sx, int(dsc.Dtype.F32))
        ov = B.dsc_tensor_from_device_ptr(ctx, o_ptr + r0 * bins * 8, n * bins * 8, 2, so, int(dsc.Dtype.C32))
        xx = B.dsc_mul(ctx, xv, xv, None)
        e_t = B.dsc_sum(ctx, xx, None, -1, True)
        B.dsc_tensor_free(ctx, xx)
        oc = B.dsc_conj(ctx, ov)
        pp = B.dsc_mul(ctx, ov, oc, None)
        B.dsc_tensor_free(ctx, oc)
        e_f = B.dsc_sum(ctx, pp, None, -1, True)
        B.dsc_tensor_free(ctx, pp)
        h_t = np.empty((n, 1), np.float32)
        h_f = np.empty((n, 1), np.complex64)
        B.dsc_copy_to_host(ctx, e_t, h_t.ctypes.data, h_t.nbytes)
        B.dsc_copy_to_host(ctx, e_f, h_f.ctypes.data, h_f.nbytes)
        edge = np.empty((n, 2), np.complex64)                 # X[:, 0] and X[:, N/2]
        for j, col in enumerate((0, bins - 1)):
            c = B.dsc_tensor_get_slice(ctx, ov, B._DscSlice(0, n, 1), B._DscSlice(col, col + 1, 1))
            tmp = np.empty((n, 1), np.complex64)
            B.dsc_copy_to_host(ctx, c, tmp.ctypes.data, tmp.nbytes)
            edge[:, j] = tmp[:, 0]
            B.dsc_tensor_free(ctx, c)
        for t in (e_t, e_f, xv, ov):
            B.dsc_tensor_free(ctx, t)
        lhs = h_t[:, 0].astype(np.float64)
        rhs = (2.0 * h_f[:, 0].real.astype(np.float64) - np.abs(edge[:, 0].astype(np.complex128)) ** 2 - np.abs(edge[:, 1].astype(np.complex128)) ** 2) / N_FFT
        worst = max(worst, float(np.max(np.abs(lhs - rhs) / lhs)))
    return {'rel_l2_vs_cpu_oracle': float(per_row.max()), 'rows_checked': len(pick), 'rows': 'first 4 + last 4 + 64 seeded random (4321 + rank)',
            'tolerance': 1e-5, 'parseval_max_rel': worst, 'parseval_rows': rows, 'parseval_tolerance': 1e-4,
            'parseval_how': 'on the device: dsc_sum(dsc_mul(x, x)) against dsc_sum(dsc_mul(X, dsc_conj(X))), f32 accumulation',
            'ok': bool(per_row.max() <= 1e-5 and worst <= 1e-4)}


def recorded_traffic(kernel, nbytes):
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE + WRITE_SIZE, corrected as
    the guide prescribes) — NOT measured in this run: counters need their own profiled passes.  Returned only for the same
    kernel at the same algorithmic size, together with where and when it was measured; otherwise (None, None)."""
    tf = os.path.join(ROOT, 'profiles', 'traffic_latest.json')
    try:
        recs = json.load(open(tf)).get('kernels', {})
        rec = recs.get(f'{kernel}@{nbytes}') or recs.get(kernel)
        if rec and rec.get('algorithmic_bytes_per_launch') == nbytes:
            return rec['hbm_bytes_per_launch'], {'measured_in_this_run': False, 'file': 'profiles/traffic_latest.json',
                                                 'from': rec.get('source'), 'date': rec.get('date'), 'method': rec.get('method')}
    except Exception:
        pass
    return None, None


def roofline_object(kernel, path, nbytes, ms, rows, workload=None):
    """The `roofline` contract: algorithmic bytes / HIP-event launch time against the 8 TB/s HBM peak."""
    achieved = nbytes / (ms * 1e-3) / 1e9
    traffic, source = recorded_traffic(kernel, nbytes)
    o = {'bound': 'hbm', 'kernel': kernel, 'kernel_path': path, 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
         'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic, 'traffic_source': source,
         'algorithmic_bytes_per_launch': nbytes, 'kernel_ms': round(ms, 4), 'rows': rows}
    if workload:
        o['workload'] = workload
    return o


def gather_phase(args, dist, world, rank, rows, barrier_sync, gpu):
    """Put the P output shards next to each other on every rank (dsc_amd/shard.py) with each requested method, timed
    (a) pipelined with the transforms — chunk i travels while chunk i+1 is computed — and (b) on its own, and PROVEN:
    every slot of the persistent [P x shard] destination is checked against its owner's checksum and sample rows.
    Backend-agnostic: RCCL on device buffers; gloo on host buffers (CPU dry run, or host staging when --share-device
    rehearses N > 1 on one GPU); 'ipc' (direct pushes over HIP IPC) whenever there is a GPU."""
    import numpy as np
    import torch
    from dsc_amd import shard
    methods = [m for m in args.gather_methods.split(',') if m]
    bins = N_FFT // 2 + 1
    on_device = gpu is not None and args.backend == 'nccl'
    if gpu is None:
        # CPU dry run: small deterministic per-rank shards, same control flow
        g_rows, row_elems, chunk_rows = 96, 130, 40
        own = torch.from_numpy((1000.0 * (rank + 1) + np.arange(g_rows, dtype=np.float32)[:, None]
                                + np.arange(row_elems, dtype=np.float32)[None, :] / 1024.0).astype(np.float32))
        methods = [m for m in methods if m not in ('ipc', 'allgather_c', 'p2p_c')]
        ctx = None
    else:
        dsc, B, ctx, x = gpu
        g_rows, row_elems, chunk_rows = rows, 2 * bins, min(args.gather_chunk_rows, rows)
        if not on_device:                                 # the library's own RCCL communicator needs one GPU per rank
            methods = [m for m in methods if m not in ('allgather_c', 'p2p_c')]
    shard_bytes = g_rows * row_elems * 4
    out = {'layout': f'dest[{world}][{g_rows}][{row_elems}] f32, rank-major = concatenation of the shards', 'shard_bytes': shard_bytes,
           'chunk_rows': chunk_rows, 'backend': args.backend, 'variants': {}}

    dev_dest = host_dest = None
    x_chunks = None
    if gpu is not None:
        import ctypes
        dev_dest = shard.DeviceDest(ctx, world, rank, g_rows, row_elems)
        chunks = shard.chunk_bounds(g_rows, chunk_rows)
        x_ptr = x._c_ptr.contents.data

        def x_view(r0, n):
            shp = (ctypes.c_int * 2)(n, N_FFT)
            return B.dsc_tensor_from_device_ptr(ctx, x_ptr + r0 * N_FFT * 4, n * N_FFT * 4, 2, shp, int(dsc.Dtype.F32))
        x_chunks = [x_view(r0, n) for r0, n in chunks]
        out_chunks = [dev_dest.own_slot_tensor(r0, n, dsc.Dtype.C32, bins) for r0, n in chunks]
        if not on_device:
            host_dest = torch.empty((world, g_rows, row_elems), dtype=torch.float32)
    else:
        host_dest = torch.empty((world, g_rows, row_elems), dtype=torch.float32)

    def run(method, with_compute):
        """One pass over the chunks: [transform chunk i ->] push chunk i; finish.  Returns seconds (max over ranks)."""
        use_dev = gpu is not None and (on_device or method == 'ipc')
        dest = dev_dest.tensor if use_dev else host_dest
        g = shard.ShardGather(dist, dest, chunk_rows, method=method, ctx=ctx, dest_ptr=dev_dest.ptr if use_dev else None)
        try:
            barrier_sync()
            t0 = time.perf_counter()
            for i, (r0, n) in enumerate(g.chunks):
                if gpu is None:
                    if with_compute:
                        dest[rank, r0:r0 + n] = own[r0:r0 + n]
                else:
                    if with_compute:
                        B.dsc_rfft(ctx, x_chunks[i], out_chunks[i], -1, -1)        # writes this rank's slot in place
                    if not use_dev and with_compute:
                        # host staging (gloo rehearsal on one GPU): chunk i leaves the device before it travels
                        B.dsc_copy_to_host(ctx, out_chunks[i], dest[rank, r0:r0 + n].data_ptr(), n * row_elems * 4)
                g.push(i)
            g.finish()
            dt = time.perf_counter() - t0
            td = torch.tensor([dt], dtype=torch.float64, device='cuda' if on_device else 'cpu')
            dist.all_reduce(td, op=dist.ReduceOp.MAX)
            return float(td.item()), g.verify()
        finally:
            g.close()

    for m in methods:
        try:
            target = dev_dest.tensor if (gpu is not None and (on_device or m == 'ipc')) else host_dest
            target.fill_(-1.0)                            # every method must fill the destination itself
            if gpu is not None:
                torch.cuda.synchronize()
            run(m, True)                                  # warm-up: communicator, IPC mappings, clocks
            target.fill_(-1.0)
            if gpu is not None:
                torch.cuda.synchronize()
            t_pipe, v = run(m, True)
            t_only, v2 = run(m, False)
            out['variants'][m] = {
                'pipelined_ms': round(t_pipe * 1e3, 3), 'ms': round(t_only * 1e3, 3),
                'recv_GBps_per_gpu': round((world - 1) * shard_bytes / t_only / 1e9, 2),
                'GBps_per_link': round(shard_bytes / t_only / 1e9, 2),      # what one peer's shard needs of one xGMI link (mesh methods: one link per peer)
                'verified': bool(v['verified'] and v2['verified']), 'rows_sampled_per_slot': v['rows_sampled_per_slot'],
                'memory': 'device' if (gpu is not None and (on_device or m == 'ipc')) else 'host (staging)' if gpu is not None else 'host',
            }
        except Exception as e:
            out['variants'][m] = {'error': repr(e)[:300], 'verified': False}
    if gpu is not None:
        for t in x_chunks:
            B.dsc_tensor_free(ctx, t)
        dev_dest.free()
    ok = [m for m in methods if out['variants'].get(m, {}).get('verified')]
    head = out['variants'].get('allgather') if 'allgather' in ok else (out['variants'][ok[0]] if ok else None)
    if head is not None:
        out.update({'ms': head['ms'], 'recv_GBps_per_gpu': head['recv_GBps_per_gpu'], 'GBps_per_link': head['GBps_per_link'], 'pipelined_ms': head['pipelined_ms'],
                    'method': 'allgather' if 'allgather' in ok else ok[0]})
    out['verified'] = bool(methods) and len(ok) == len(methods)
    out['note'] = ('separate phase, not included in value.  ms = exchange alone (shards already computed); pipelined_ms = transforms '
                   'of all chunks + exchange, chunk i travelling while chunk i+1 is transformed; verified = every slot of every '
                   "rank's destination equals its owner's shard (checksum + sample rows, bit exact)")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--ramp-ms', type=float, default=100.0,
                    help='untimed launches of the same step for this long before the warm-up steps: the GPU clocks need tens of '
                         'milliseconds of load to come back up after the host-side parity check (0 disables)')
    ap.add_argument('--batch', type=int, default=BATCH, help='rows per GPU (default: BASELINE config 2)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-kernels', action='store_true', help='skip the untimed-for-value irfft / fused-filter measurements')
    ap.add_argument('--no-c5', action='store_true', help='skip the config-5 (f64 N=262144) entry of other_kernels')
    ap.add_argument('--chunks', type=int, default=8, help='N = 1 only: also time the rows of configs[3] as this many sequential 8192-row launches on one GPU (0 disables)')
    ap.add_argument('--no-allgather', action='store_true')
    ap.add_argument('--allgather-timeout', type=float, default=240.0, help='seconds the separate shard-reassembly phase may take before it is abandoned (every rank then exits 3)')
    ap.add_argument('--gather-methods', default='allgather,p2p,ipc,allgather_c', help='exchange methods of dsc_amd/shard.py to run and verify, in this order')
    ap.add_argument('--gather-chunk-rows', type=int, default=1024, help='rows per exchange step of the chunked methods')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (gloo for the CPU dry run)')
    ap.add_argument('--dry-run', action='store_true', help='exercise the multi-process plumbing without a GPU (tests)')
    ap.add_argument('--share-device', action='store_true', help='testing aid: every rank uses device 0 (rehearse N > 1 on a one-GPU box; use with --backend gloo)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f'--gpus {args.gpus} needs a launcher: python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...')
        args.gpus = world

    import numpy as np
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.share_device:
            local_rank = 0
        if not args.dry_run:
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world)

    def barrier_sync():
        if dist is not None:
            dist.barrier()
        if not args.dry_run:
            torch.cuda.synchronize()

    rows = args.batch
    extra = {}
    if args.dry_run:
        # same control flow, no device: the "step" is a fixed amount of host work
        def step():
            time.sleep(0.001)
        kernel_ms = None
        parity = None
        path = 'dry-run'
    else:
        with c_stdout_to_stderr():
            import dsc_amd as dsc
            from dsc_amd import _bindings as B
            from dsc_amd.context import _get_ctx
            # main arena: x + out, irfft / filter buffers, config 5 (4 + 4 GiB); scratch: the two-pass intermediate in 512-row chunks
            dsc.init(2 * rows * BYTES_PER_ROW + ((14 << 30) if not args.no_other_kernels else (1 << 30)), (2 << 30) if not args.no_other_kernels else (1 << 30), device=local_rank)
        ctx = _get_ctx()
        x_host = synthetic_rows(rows, rank)
        x = dsc.from_numpy(x_host)
        out = dsc.empty((rows, N_FFT // 2 + 1), dsc.Dtype.C32)

        def step():
            B.dsc_rfft(ctx, x._c_ptr, out._c_ptr, -1, -1)

        step()
        path = dsc.last_fft_path()
        # parity of this very buffer (checker only; untimed): sample rows against the CPU oracle, Parseval on all rows on the device
        with c_stdout_to_stderr():
            parity = device_parity(dsc, B, ctx, x, out, x_host, rows, rank)

    if not args.dry_run and args.ramp_ms > 0:
        # clock ramp (untimed, before the W warm-up steps): the parity check above left the GPU idle
        t_ramp = time.perf_counter()
        while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
            for _ in range(10):
                step()
            dsc.synchronize()
    for _ in range(args.warmup):
        step()
    barrier_sync()
    marks = None
    if not args.dry_run:
        # one HIP event per step boundary on the stream the kernel is launched on (recorded, never waited for inside the loop)
        cstream = torch.cuda.ExternalStream(B.dsc_stream(ctx))
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        B.dsc_timer_start(ctx)          # HIP events on the stream the kernel is launched on
        marks[0].record(cstream)
    t0 = time.perf_counter()
    host_marks = [t0]
    for i in range(args.steps):
        step()
        if marks is not None:
            marks[i + 1].record(cstream)
        else:
            host_marks.append(time.perf_counter())      # dry run: the host clock stands in for the events
    if not args.dry_run:
        kernel_ms = B.dsc_timer_stop(ctx) / args.steps
    barrier_sync()
    elapsed = time.perf_counter() - t0
    per_step = (sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)) if marks is not None
                else sorted((host_marks[i + 1] - host_marks[i]) * 1e3 for i in range(args.steps)))

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if args.dry_run or args.backend == 'gloo' else 'cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- the other kernels of the hot path (configs[1] irfft on the same buffers, configs[2] fused filter on half the batch,
    # configs[4] rfft f64 N=262144): each a full roofline object, reported next to the metric, never part of `value`
    if not args.dry_run and not args.no_other_kernels:
        try:
            def timed(f, n=max(args.steps, 50)):
                # the host-side setup before each of these kernels (uploads of a GiB) leaves the GPU idle and its clocks low: the
                # same untimed ramp as the metric's loop, then max(W, 10) warm-up launches
                t_end = time.perf_counter() + args.ramp_ms / 1e3
                while time.perf_counter() < t_end:
                    for _ in range(10):
                        f()
                    dsc.synchronize()
                for _ in range(max(10, args.warmup)):
                    f()
                dsc.synchronize()
                B.dsc_timer_start(ctx)
                for _ in range(n):
                    f()
                launches.append(n)
                return B.dsc_timer_stop(ctx) / n

            ok = {}
            launches = []
            back = dsc.empty((rows, N_FFT), dsc.Dtype.F32)
            ms_i = timed(lambda: B.dsc_irfft(ctx, out._c_ptr, back._c_ptr, -1, -1))
            ok['irfft'] = roofline_object('irfft64k_kernel', dsc.last_fft_path(), rows * BYTES_PER_ROW, ms_i, rows,
                                          f'1-D irfft f32 N={N_FFT} batch={rows} (BASELINE configs[1]), input = the spectrum the timed rfft wrote')
            ok['irfft']['launches'] = launches[-1]
            del back
            rngf = np.random.default_rng(7)
            H = dsc.from_numpy((rngf.standard_normal(N_FFT // 2 + 1) + 1j * rngf.standard_normal(N_FFT // 2 + 1)).astype(np.complex64))
            half = rows // 2 if rows >= 2 else rows
            s_half = dsc.from_numpy(x_host[:half])
            y_half = dsc.empty((half, N_FFT), dsc.Dtype.F32)
            ms_f = timed(lambda: B.dsc_filter_fft(ctx, s_half._c_ptr, H._c_ptr, y_half._c_ptr))
            ok['fused_filter'] = roofline_object('filter64k_kernel', dsc.last_fft_path(), half * N_FFT * 8, ms_f, half,
                                                 f'y = irfft(rfft(s) * H) fused, N={N_FFT} batch={half} (BASELINE configs[2]), 4 B in + 4 B out per sample, H [32769] c32 broadcast')
            ok['fused_filter']['launches'] = launches[-1]
            del H, s_half, y_half
            if rows >= BATCH and not args.no_c5:
                n5, b5 = 262144, 2048
                x5 = dsc.empty((b5, n5), dsc.Dtype.F64)
                X5 = dsc.empty((b5, n5 // 2 + 1), dsc.Dtype.C64)
                blk = np.random.default_rng(99).standard_normal((64, n5))
                import ctypes
                shp = (ctypes.c_int * 2)(64, n5)
                for r0 in range(0, b5, 64):               # the same 64 random rows 32 times: no 4 GiB host array
                    piece = B.dsc_tensor_from_device_ptr(ctx, x5._c_ptr.contents.data + r0 * n5 * 8, 64 * n5 * 8, 2, shp, int(dsc.Dtype.F64))
                    B.dsc_copy_from_host(ctx, piece, blk.ctypes.data, blk.nbytes)
                    B.dsc_tensor_free(ctx, piece)
                ms_5 = timed(lambda: B.dsc_rfft(ctx, x5._c_ptr, X5._c_ptr, -1, -1), n=20)
                p5 = dsc.last_fft_path()
                k5 = 'fused_l2_kernel' if p5.endswith('fused_l2') else 'two_pass_rows_kernel + two_pass_cols_kernel'
                ok['rfft_f64_262144'] = roofline_object(k5, p5, b5 * (n5 * 8 + (n5 // 2 + 1) * 16), ms_5, b5,
                                                        f'1-D rfft f64 N={n5} batch={b5} (BASELINE configs[4])' +
                                                        ('; one launch, the four-step intermediate stays in the XCD-local L2' if k5 == 'fused_l2_kernel'
                                                         else '; two kernels, the time is their sum'))
                ok['rfft_f64_262144']['launches'] = launches[-1]
                del x5, X5
            ok['note'] = 'same process, same HIP-event timer, this rank only, random inputs; not included in value'
            extra['other_kernels'] = ok
        except Exception as e:
            extra['other_kernels'] = {'error': repr(e)[:200]}

    # ---- N = 1 only: BASELINE configs[3]'s 65536 rows on ONE GPU as 8 sequential 8192-row launches (the 1-GPU
    # comparison point of the 8-GPU line; a single 65536 x 65536 tensor cannot exist, `int ne`, dsc.h:104)
    if not args.dry_run and world == 1 and args.chunks > 0 and rows == BATCH:
        try:
            import ctypes
            bins = N_FFT // 2 + 1
            ins, outs, raw = [], [], []
            shp_i, shp_o = (ctypes.c_int * 2)(rows, N_FFT), (ctypes.c_int * 2)(rows, bins)
            for c in range(args.chunks):
                pi, po = B.dsc_device_alloc(ctx, rows * N_FFT * 4), B.dsc_device_alloc(ctx, rows * bins * 8)
                if not pi or not po:
                    raise MemoryError(f'chunk {c}: dsc_device_alloc failed')
                raw += [pi, po]
                ti = B.dsc_tensor_from_device_ptr(ctx, pi, rows * N_FFT * 4, 2, shp_i, int(dsc.Dtype.F32))
                B.dsc_copy_from_host(ctx, ti, x_host.ctypes.data, x_host.nbytes)
                ins.append(ti)
                outs.append(B.dsc_tensor_from_device_ptr(ctx, po, rows * bins * 8, 2, shp_o, int(dsc.Dtype.C32)))

            def all_chunks():
                for ti, to in zip(ins, outs):
                    B.dsc_rfft(ctx, ti, to, -1, -1)
            t_end = time.perf_counter() + args.ramp_ms / 1e3
            while time.perf_counter() < t_end:
                all_chunks()
                dsc.synchronize()
            for _ in range(3):
                all_chunks()
            dsc.synchronize()
            reps = max(5, args.steps // args.chunks)
            B.dsc_timer_start(ctx)
            for _ in range(reps):
                all_chunks()
            ms_c4 = B.dsc_timer_stop(ctx) / reps
            extra['config4_on_one_gpu'] = {
                'workload': f'1-D rfft f32 N={N_FFT}, {args.chunks * rows} rows as {args.chunks} sequential launches of {rows} rows, every chunk with its own '
                            f'input and output buffers resident in HBM ({args.chunks * rows * BYTES_PER_ROW / 2**30:.0f} GiB)',
                'rows': args.chunks * rows, 'chunks': args.chunks, 'ms': round(ms_c4, 4), 'ms_per_chunk': round(ms_c4 / args.chunks, 4),
                'value': round(args.chunks * rows * N_FFT / ms_c4 / 1e6, 3), 'unit': 'GSamples/s', 'repetitions': reps,
                'frac_of_hbm_peak': round(args.chunks * rows * BYTES_PER_ROW / (ms_c4 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                'data': 'every chunk holds the rank-0 shard (seed 1234); timing does not depend on the values',
                'note': 'the N = 1 denominator for configs[3] (8 x MI355X, one 8192-row shard each): same rows, same launches, one GPU'}
            for t in ins + outs:
                B.dsc_tensor_free(ctx, t)
            for ptr in raw:
                B.dsc_device_free(ctx, ptr)
        except Exception as e:
            extra['config4_on_one_gpu'] = {'error': repr(e)[:200]}

    if rank == 0:
        total_samples = world * rows * N_FFT * args.steps
        ms_per_step = elapsed / args.steps * 1e3
        line = {
            'metric': 'batched 1-D rfft GSamples/s',
            'value': round(total_samples / elapsed / 1e9, 3),
            'unit': 'GSamples/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 4),
            'min_ms': round(per_step[0], 4) if per_step else None,
            'median_ms': round(per_step[len(per_step) // 2], 4) if per_step else None,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': f'1-D rfft f32 N={N_FFT} batch={rows} per GPU (BASELINE configs[1]), '
                                   f'inputs resident in HBM, standard_normal seed 1234+rank',
                       'n_fft': N_FFT, 'batch_per_gpu': rows, 'global_batch': world * rows, 'parallelism': f'batch-shard x{world}',
                       'kernel_path': path, 'clock_ramp_ms_untimed': 0.0 if args.dry_run else args.ramp_ms},
        }
        if not args.dry_run:
            line['roofline'] = roofline_object('rfft64k_kernel' if path == 'r2c_64k_regs' else path, path, rows * BYTES_PER_ROW, kernel_ms, rows)
            line['parity'] = parity
            if world == 1 and not args.no_cpu_baseline:
                line['cpu_baseline'] = cpu_baseline()
                line['cpu_baseline']['gpu_over_cpu'] = round(line['value'] / line['cpu_baseline']['value'], 1)
    else:
        line = None

    # A collective that never returns must not cost the measurement above: past the deadline rank 0 prints the line it
    # has (with the reason in `allgather`) and EVERY rank exits non-zero, so that the hang is visible to the launcher.
    printed = [False]

    def deadline(seconds, what):
        def bail():
            if rank == 0 and not printed[0]:
                line.update(extra)
                line.setdefault('allgather', {})['error'] = f'{what} did not finish within {seconds:.0f} s; abandoned'
                print(json.dumps(line), flush=True)
            os._exit(3)
        t = threading.Timer(seconds, bail)
        t.daemon = True
        t.start()
        return t

    # ---- reassembly of the output shards: its own phase, never part of `value`
    if dist is not None and not args.no_allgather:
        guard = deadline(args.allgather_timeout, 'shard reassembly phase')
        try:
            extra['allgather'] = gather_phase(args, dist, world, rank, rows, barrier_sync,
                                              None if args.dry_run else (dsc, B, ctx, x))
        except Exception as e:       # the metric must survive a collective problem
            extra['allgather'] = {'error': repr(e)[:300]}
        guard.cancel()

    if rank == 0:
        line.update(extra)
        print(json.dumps(line), flush=True)
        printed[0] = True

    if dist is not None:
        guard = deadline(120.0, 'process-group shutdown')
        dist.barrier()
        dist.destroy_process_group()
        guard.cancel()


if __name__ == '__main__':
    main()
