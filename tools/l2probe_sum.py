"""Merge tools/l2probe's timing run with its two PMC passes (FETCH_SIZE x2, WRITE_SIZE x1, KiB: the calibration of
tools/calib_copy.hip): per variant, how much of the scratch traffic reached the L2 <-> fabric boundary."""
import csv, glob, re, sys

out = sys.argv[1]
GB = 4 * 2 ** 30 / 1e9                     # in = out = scratch = 4 GiB per launch
lines = [l.rstrip('\n') for l in open(f'{out}/timing.txt') if l.startswith('variant') or l.startswith('#')]


def counters(name, factor):
    rows = []
    for f in glob.glob(f'{out}/pmc_{name}/**/*counter_collection.csv', recursive=True):
        for n, r in enumerate(csv.DictReader(open(f))):
            if ('probe<' in r['Kernel_Name'] or 'probe_split<' in r['Kernel_Name'] or 'probe_pipe<' in r['Kernel_Name']) and r['Counter_Name'] == name:
                rows.append((int(r.get('Dispatch_Id', n)), float(r['Counter_Value']) * factor * 1024 / 1e9))
    return [v for _, v in sorted(rows)]


fetch, write = counters('FETCH_SIZE', 2.0), counters('WRITE_SIZE', 1.0)
variants = [l for l in lines if l.startswith('variant')]
print(f'# {len(variants)} variants, {len(fetch)} / {len(write)} profiled dispatches; in = out = scratch = {GB:.3f} GB per launch')
print('# wb = (written - out) / scratch: share of the scratch lines written back;  rf = (fetched - in) / scratch: share fetched again')
i = 0
for l in lines:
    if l.startswith('#'):
        print(l)
        continue
    f = fetch[i] if i < len(fetch) else float('nan')
    w = write[i] if i < len(write) else float('nan')
    no_in, no_out = ' NO-IN' in l, ' NO-OUT' in l
    fr = re.search(r'frac +(\d+)/16', l)
    us = re.search(r'([0-9.]+ us per slot cycle)', l)
    A = GB * (int(fr.group(1)) / 16 if fr else 1.0)
    wb = (w - (0 if no_out else GB)) / A
    rf = (f - (0 if no_in else GB)) / A
    ms = re.search(r'([0-9.]+) ms', l).group(1)
    head = l.split(' delay')[0].replace(' refresh 0', '')
    ms = re.search(r'([0-9.]+) ms', l.split(' delay')[1]).group(1)
    print(f'{head} | {ms:>8} ms | fetched {f:6.2f} GB written {w:6.2f} GB | wb {wb:5.2f} rf {rf:5.2f}' + (f' | {us.group(1)}' if us else ''))
    i += 1
