#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh output directory into summary.md + traffic.json
(the files that get committed under profiles/)."""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
KEYS = ('rfft64k', 'irfft64k', 'filter64k')


def counter(name, sub, match):
    vals = []
    for f in glob.glob(f'{out}/{sub}/*/*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == name and any(k in r['Kernel_Name'] for k in match):
                vals.append(float(r['Counter_Value']))
    return vals


lines = ['# rocprofv3 summary (' + os.path.basename(out.rstrip('/')) + ')', '']
bench = json.load(open(f'{out}/bench.json'))
lines += ['## bench.py line (un-profiled run)', '', '```json', json.dumps(bench), '```', '']
lines += ['## `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-other-kernels` (plus the untimed 100 ms clock ramp, ≈110 launches)', '']
for f in glob.glob(f'{out}/trace/*/*_kernel_stats.csv'):
    lines += ['```csv'] + open(f).read().strip().splitlines() + ['```', '']
avg_ns = None
for f in glob.glob(f'{out}/trace/*/*_kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if any(k in r['Name'] for k in KEYS):
            avg_ns = float(r['AverageNs'])
KB = 1024.0
cal_f = counter('FETCH_SIZE', 'cal_fetch', ('calib',))
cal_w = counter('WRITE_SIZE', 'cal_write', ('calib',))
known = 8192 * 262144
f_corr = known / (sum(cal_f) / len(cal_f) * KB) if cal_f else None
w_corr = known / (sum(cal_w) / len(cal_w) * KB) if cal_w else None
fs = counter('FETCH_SIZE', 'pmc_fetch', KEYS)
ws = counter('WRITE_SIZE', 'pmc_write', KEYS)
lines += ['## HBM traffic (PMC, separate passes)', '']
lines += [f'- calibration kernel (tools/calib_copy.hip, same access widths, {known} B read and written per launch): '
          f'FETCH_SIZE reads {sum(cal_f)/len(cal_f):.1f} KB -> correction x{f_corr:.4f}; '
          f'WRITE_SIZE reads {sum(cal_w)/len(cal_w):.1f} KB -> correction x{w_corr:.4f}' if cal_f and cal_w else '- calibration missing']
traffic = None
if fs and ws and f_corr and w_corr:
    fetch_b = sum(fs) / len(fs) * KB * f_corr
    write_b = sum(ws) / len(ws) * KB * w_corr
    traffic = fetch_b + write_b
    alg = bench['roofline']['algorithmic_bytes_per_launch']
    lines += [f'- rfft kernel: FETCH_SIZE {sum(fs)/len(fs):.1f} KB raw -> {fetch_b/1e9:.4f} GB corrected; '
              f'WRITE_SIZE {sum(ws)/len(ws):.1f} KB raw -> {write_b/1e9:.4f} GB corrected',
              f'- HBM bytes per launch {traffic/1e9:.4f} GB vs algorithmic {alg/1e9:.4f} GB: x{traffic/alg:.3f}']
if avg_ns:
    alg = bench['roofline']['algorithmic_bytes_per_launch']
    lines += ['', f'- rocprof average kernel duration {avg_ns/1e6:.4f} ms -> {alg/avg_ns:.1f} GB/s algorithmic '
              f'({alg/avg_ns/8000*100:.1f} % of 8 TB/s); bench.py HIP-event figure: {bench["roofline"]["kernel_ms"]} ms']
open(f'{out}/summary.md', 'w').write('\n'.join(lines) + '\n')
json.dump({'hbm_bytes_per_launch': traffic, 'rocprof_avg_kernel_ns': avg_ns, 'fetch_correction': f_corr, 'write_correction': w_corr,
           'source': out}, open(f'{out}/traffic.json', 'w'))
print('\n'.join(lines[-8:]))
