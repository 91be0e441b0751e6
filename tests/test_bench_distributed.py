"""N > 1 plumbing of bench.py on CPU: two gloo ranks launched exactly as the driver launches
the GPU run (torch.distributed.run, one process per device), with --dry-run standing in for
the device work.  Checks rendezvous, barrier, max-over-ranks timing and the single JSON line."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_two_rank_dry_run():
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1',
           '--backend', 'gloo', '--dry-run']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['steps'] == 4 and j['warmup'] == 1
    assert j['scaling'] == 'weak' and j['higher_is_better'] is True and j['unit'] == 'GSamples/s'
    assert j['config']['global_batch'] == 2 * 8192
    # 4 steps of >= 1 ms each on the slowest rank
    assert j['ms_per_step'] >= 1.0
    assert 1.0 <= j['min_ms'] <= j['median_ms']                       # per-step times (reference convention: the minimum)
    assert abs(j['value'] - 2 * 8192 * 65536 / (j['ms_per_step'] * 1e-3) / 1e9) / j['value'] < 1e-2
    # the reassembly phase ran for real on two gloo ranks (deterministic per-rank arrays) and proved itself
    g = j['allgather']
    assert g['verified'] is True and g['ms'] > 0 and g['recv_GBps_per_gpu'] > 0 and g['GBps_per_link'] > 0
    assert set(g['variants']) == {'allgather', 'p2p'}              # 'ipc' needs device memory
    assert all(v['verified'] is True for v in g['variants'].values())


def test_single_process_requires_launcher():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run'],
                       capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode != 0 and 'torch.distributed.run' in r.stderr
