#!/usr/bin/env python3
"""Condense a tools/profile_families.sh output directory into one markdown summary (committed under profiles/):
per case the HIP-event line, the rocprofv3 kernel stats of the kernels that carry the case, and — where PMC passes
were taken — every counter averaged over the dispatches of the dominant kernel."""
import csv
import glob
import json
import os
import sys
import time

out = sys.argv[1].rstrip('/')
KB = 1024.0


def short_name(full):
    """'void (anonymous namespace)::fft_mid_kernel<float, 4, ...>(args)' -> 'fft_mid_kernel<float, 4, ...>'"""
    n = full.replace('(anonymous namespace)::', '')
    if n.startswith('void '):
        n = n[5:]
    depth = 0
    for i, ch in enumerate(n):            # cut the argument list: the first '(' outside template brackets
        if ch == '<':
            depth += 1
        elif ch == '>':
            depth -= 1
        elif ch == '(' and depth == 0:
            n = n[:i]
            break
    return n[:90]


def cal(sub):
    vals = []
    for f in glob.glob(f'{out}/{sub}/*/*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'calib' in r['Kernel_Name']:
                vals.append(float(r['Counter_Value']))
    return sum(vals) / len(vals) if vals else None


known = 8192 * 262144
cf, cw = cal('cal_fetch'), cal('cal_write')
f_corr = known / (cf * KB) if cf else None
w_corr = known / (cw * KB) if cw else None
print(f'# rocprofv3 per-family summary ({os.path.basename(out)})\n')
print('Every case: `python3 tools/run_op.py <case>` (30 timed launches x 3 after a warm-up; HIP events on the context stream), then the same '
      'command under `rocprofv3 --kernel-trace --stats`.  % = algorithmic bytes (SURVEY 8d) / average duration / 8 TB/s.\n')
if f_corr:
    print(f'PMC calibration (tools/calib_copy.hip, {known} B read + written per launch): FETCH_SIZE x{f_corr:.4f}, WRITE_SIZE x{w_corr:.4f}\n')
print('| case | path | HIP-event ms | % of 8 TB/s | rocprof: kernel (calls) avg ms | sum of avg ms | % from rocprof |')
print('|---|---|---|---|---|---|---|')
dominant = {}
for d in sorted(glob.glob(f'{out}/*/plain.json')):
    c = os.path.basename(os.path.dirname(d))
    try:
        j = json.loads(open(d).read().strip().splitlines()[-1])
    except Exception:
        continue
    ks = os.path.join(os.path.dirname(d), 'kernel_stats.csv')
    kern = []
    if os.path.exists(ks):
        rows = list(csv.DictReader(open(ks)))
        # kernels of the timed loop: those called at least as often as the timed launches (3 x 30), set-up kernels are called once or twice
        for r in rows:
            if int(r['Calls']) >= 90:
                kern.append((short_name(r['Name']), int(r['Calls']), float(r['AverageNs']) / 1e6))
    tot = sum(k[2] * k[1] for k in kern) / max(1, max((k[1] for k in kern), default=1))
    # one kernel NAME launched several times per operator call (both passes of the complex four-step along an axis at n1 = n2):
    # the per-name average counts once per launch, the operator's time is that many launches
    if len(kern) == 1 and tot > 0 and j['hip_event_ms'] / tot >= 1.7:
        tot *= round(j['hip_event_ms'] / tot)
    if kern:
        dominant[c] = max(kern, key=lambda k: k[2])[0]
    ktxt = '; '.join(f'`{k[0]}` ({k[1]}) {k[2]:.4f}' for k in kern) or 'n/a'
    pct = j['algorithmic_bytes'] / tot / 1e6 / 80 if tot else 0
    print(f"| {c} | {j['path']} | {j['hip_event_ms']:.4f} | {j['frac_of_8TBps'] * 100:.1f} | {ktxt} | {tot:.4f} | {pct:.1f} |")
print()
traffic = {}
for d in sorted(glob.glob(f'{out}/*/pmc1')):
    cdir = os.path.dirname(d)
    c = os.path.basename(cdir)
    acc = {}
    meta = None
    key = dominant.get(c, '')
    for f in glob.glob(f'{cdir}/pmc*/*/*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if key and key.split('<')[0] not in r['Kernel_Name'].replace('(anonymous namespace)::', ''):
                continue
            acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
            meta = (r['Grid_Size'], r['Workgroup_Size'], r['LDS_Block_Size'], r['VGPR_Count'], r.get('Accum_VGPR_Count', '?'), r['SGPR_Count'])
    if not acc:
        continue
    j = json.loads(open(f'{cdir}/plain.json').read().strip().splitlines()[-1])
    print(f'## PMC: {c} (kernel `{key}`; grid, wg, lds, vgpr, agpr, sgpr = {meta})\n')
    print('| counter | average per dispatch |')
    print('|---|---|')
    for k in sorted(acc):
        v = acc[k]
        print(f'| {k} | {sum(v) / len(v):.1f} (n={len(v)}) |')
    if 'FETCH_SIZE' in acc and 'WRITE_SIZE' in acc and f_corr:
        fb = sum(acc['FETCH_SIZE']) / len(acc['FETCH_SIZE']) * KB * f_corr
        wb = sum(acc['WRITE_SIZE']) / len(acc['WRITE_SIZE']) * KB * w_corr
        alg = j['algorithmic_bytes']
        print(f'\nHBM bytes per launch: fetch {fb / 1e9:.4f} GB + write {wb / 1e9:.4f} GB = {(fb + wb) / 1e9:.4f} GB vs algorithmic {alg / 1e9:.4f} GB: x{(fb + wb) / alg:.3f}')
        # one record per (kernel, size): template kernels serve several sizes (bench.py looks up `name@bytes` first)
        traffic[key.split('<')[0]] = traffic[f"{key.split('<')[0]}@{alg}"] = {'hbm_bytes_per_launch': fb + wb, 'algorithmic_bytes_per_launch': alg, 'ratio': round((fb + wb) / alg, 4),
                                      'source': f'{out}/{c} (tools/profile_families.sh)', 'date': time.strftime('%Y-%m-%d'),
                                      'method': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, corrected x%.4f / x%.4f on tools/calib_copy.hip' % (f_corr, w_corr)}
    if 'SQ_WAVE_CYCLES' in acc:
        wc = sum(acc['SQ_WAVE_CYCLES']) / len(acc['SQ_WAVE_CYCLES'])
        parts = []
        for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_LDS'):
            if k in acc:
                parts.append(f'{k} {sum(acc[k]) / len(acc[k]) / wc * 100:.1f} %')
        print('\nShare of SQ_WAVE_CYCLES: ' + ', '.join(parts))
    if 'SQ_LDS_BANK_CONFLICT' in acc and 'SQ_LDS_IDX_ACTIVE' in acc:
        print(f"\nLDS bank-conflict cycles / LDS active cycles: {sum(acc['SQ_LDS_BANK_CONFLICT']) / max(1.0, sum(acc['SQ_LDS_IDX_ACTIVE'])) * 100:.1f} %")
    print()

if traffic:
    json.dump({'kernels': traffic}, open(f'{out}/traffic.json', 'w'), indent=1)
