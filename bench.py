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
  roofline      dominant kernel vs the 8 TB/s HBM peak (algorithmic bytes / HIP-event time)
  cpu_baseline  the reference's own CPU code (oracle/_ref, kind "reference") or our C
                restatement (kind "port") timed on this host, rank 0, N == 1 only
  parity        rel-L2 of a few GPU rows against the CPU oracle (checker, not timed)
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
    ap.add_argument('--no-allgather', action='store_true')
    ap.add_argument('--allgather-timeout', type=float, default=180.0, help='seconds the separate RCCL all-gather phase may take before it is abandoned')
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
            dsc.init(2 * rows * BYTES_PER_ROW + (1 << 30), 1 << 30, device=local_rank)
        ctx = _get_ctx()
        x_host = synthetic_rows(rows, rank)
        x = dsc.from_numpy(x_host)
        out = dsc.empty((rows, N_FFT // 2 + 1), dsc.Dtype.C32)

        def step():
            B.dsc_rfft(ctx, x._c_ptr, out._c_ptr, -1, -1)

        step()
        path = dsc.last_fft_path()
        # parity of this very buffer against the CPU oracle (checker only; untimed)
        from oracle import port
        first = np.empty((4, N_FFT // 2 + 1), np.complex64)
        B.dsc_copy_to_host(ctx, out._c_ptr, first.ctypes.data, first.nbytes)
        want = port.rfft(x_host[:4])
        parity = float(np.linalg.norm(first - want) / np.linalg.norm(want))

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
    if not args.dry_run:
        B.dsc_timer_start(ctx)          # HIP events on the stream the kernel is launched on
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if not args.dry_run:
        kernel_ms = B.dsc_timer_stop(ctx) / args.steps
    barrier_sync()
    elapsed = time.perf_counter() - t0

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if args.dry_run or args.backend == 'gloo' else 'cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- the other two kernels of the hot path on the same buffers (configs[1] irfft, configs[2] fused filter on half the
    # batch): reported next to the metric, never part of `value`
    if not args.dry_run and not args.no_other_kernels:
        try:
            back = dsc.empty((rows, N_FFT), dsc.Dtype.F32)
            H = dsc.from_numpy(np.ones(N_FFT // 2 + 1, np.complex64))
            half = rows // 2 if rows >= 2 else rows
            s_half = dsc.from_numpy(x_host[:half])
            y_half = dsc.empty((half, N_FFT), dsc.Dtype.F32)

            def timed(f, n=args.steps):
                for _ in range(max(10, args.warmup)):
                    f()
                dsc.synchronize()
                B.dsc_timer_start(ctx)
                for _ in range(n):
                    f()
                return B.dsc_timer_stop(ctx) / n

            ms_i = timed(lambda: B.dsc_irfft(ctx, out._c_ptr, back._c_ptr, -1, -1))
            p_i = dsc.last_fft_path()
            ms_f = timed(lambda: B.dsc_filter_fft(ctx, s_half._c_ptr, H._c_ptr, y_half._c_ptr))
            p_f = dsc.last_fft_path()
            extra['other_kernels'] = {
                'irfft': {'ms': round(ms_i, 4), 'frac_of_8TBps': round(rows * BYTES_PER_ROW / (ms_i * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          'rows': rows, 'kernel_path': p_i},
                'fused_filter': {'ms': round(ms_f, 4), 'frac_of_8TBps': round(half * N_FFT * 8 / (ms_f * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 'rows': half, 'kernel_path': p_f, 'bytes_per_sample': 8},
                'note': 'same process, same HIP-event timer, this rank only; not included in value',
            }
            del back, H, s_half, y_half
        except Exception as e:
            extra['other_kernels'] = {'error': repr(e)[:200]}

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
            achieved = rows * BYTES_PER_ROW / (kernel_ms * 1e-3) / 1e9
            traffic = None
            tf = os.path.join(ROOT, 'profiles', 'traffic_latest.json')
            if os.path.exists(tf):
                try:
                    traffic = json.load(open(tf)).get('hbm_bytes_per_launch')
                except Exception:
                    traffic = None
            line['roofline'] = {
                'bound': 'hbm',
                'kernel': 'rfft64k_kernel' if path == 'r2c_64k_regs' else path,
                'achieved': round(achieved, 1),
                'peak': HBM_PEAK_GBS,
                'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBS, 4),
                'traffic': traffic,
                'algorithmic_bytes_per_launch': rows * BYTES_PER_ROW,
                'kernel_ms': round(kernel_ms, 4),
            }
            line['parity'] = {'rel_l2_vs_cpu_oracle': parity, 'rows_checked': 4, 'tolerance': 1e-5}
            if world == 1 and not args.no_cpu_baseline:
                line['cpu_baseline'] = cpu_baseline()
                line['cpu_baseline']['gpu_over_cpu'] = round(line['value'] / line['cpu_baseline']['value'], 1)
    else:
        line = None

    # A collective that never returns must not cost the measurement above: past the deadline every rank leaves
    # (rank 0 prints the line first, with the reason in `allgather`).
    def deadline(seconds, what):
        def bail():
            if rank == 0 and not printed[0]:
                line['allgather'] = {'error': f'{what} did not finish within {seconds:.0f} s; skipped'}
                print(json.dumps(line), flush=True)
            os._exit(0)
        t = threading.Timer(seconds, bail)
        t.daemon = True
        t.start()
        return t

    printed = [False]
    # ---- all-gather of the output shards over xGMI: its own phase, never part of `value`
    if dist is not None and not args.dry_run and not args.no_allgather and args.backend == 'nccl':
        guard = deadline(args.allgather_timeout, 'RCCL all-gather phase')
        try:
            chunk_rows = 1024
            bins = N_FFT // 2 + 1

            class _DevView:          # zero-copy torch view of arena memory
                def __init__(self, ptr, n_f32):
                    self.__cuda_array_interface__ = {'shape': (n_f32,), 'typestr': '<f4', 'data': (ptr, False), 'version': 2}

            base = out._c_ptr.contents.data
            recv = torch.empty((world, chunk_rows * bins * 2), dtype=torch.float32, device='cuda')
            views = [torch.as_tensor(_DevView(base + c * bins * 8, chunk_rows * bins * 2), device='cuda')
                     for c in range(0, rows, chunk_rows)]
            dist.all_gather_into_tensor(recv, views[0])       # warm-up: communicator + buffers
            barrier_sync()
            g0 = time.perf_counter()
            for v in views:
                dist.all_gather_into_tensor(recv, v)
            barrier_sync()
            g = time.perf_counter() - g0
            tg = torch.tensor([g], dtype=torch.float64, device='cuda')
            dist.all_reduce(tg, op=dist.ReduceOp.MAX)
            g = float(tg.item())
            shard_bytes = rows * bins * 8
            extra['allgather'] = {
                'ms': round(g * 1e3, 3),
                'recv_GBps_per_gpu': round((world - 1) * shard_bytes / g / 1e9, 1),
                'shard_bytes': shard_bytes,
                'note': 'RCCL all_gather_into_tensor of every c32 shard, 1024-row chunks into a reused buffer; '
                        'separate phase, not included in value',
            }
        except Exception as e:       # the metric must survive a collective problem
            extra['allgather'] = {'error': repr(e)[:200]}
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
