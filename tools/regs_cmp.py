"""tools/regs_cmp.py <file.hip> "<flags A>" "<flags B>" — VGPR / spill counts per kernel for two builds of one source, differences only."""
import subprocess, sys, re, os, tempfile
def regs(src, flags):
    d = tempfile.mkdtemp()
    out = os.path.join(d, 'k.s')
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-std=c++17', '-O3', '-I/root/repo/include', '-ffp-contract=fast',
                    '-fno-slp-vectorize', '-S', '--cuda-device-only', src, '-o', out] + flags.split(), check=True, stderr=subprocess.DEVNULL)
    r, name, cur = {}, None, {}
    for line in open(out):
        m = re.match(r'\s+\.name:\s+(\S+)', line)
        if m and m.group(1).startswith('_Z'): name = m.group(1)
        m = re.match(r'\s+\.(vgpr_count|vgpr_spill_count|agpr_count):\s+(\d+)', line)
        if m: cur[m.group(1)] = int(m.group(2))
        if line.strip().startswith('.wavefront_size') and name:
            r[name] = dict(cur); cur = {}; name = None
    return r
src, fa, fb = sys.argv[1], sys.argv[2], sys.argv[3]
a, b = regs(src, fa), regs(src, fb)
names = subprocess.run(['c++filt'], input='\n'.join(a), capture_output=True, text=True).stdout.split('\n')
worse = 0
for n, dn in zip(a, names):
    x, y = a[n], b.get(n, {})
    if x != y:
        flag = ' <-- SPILLS' if y.get('vgpr_spill_count', 0) > x.get('vgpr_spill_count', 0) else ''
        worse += bool(flag)
        print(f"{dn[:150]}: {x.get('vgpr_count')}/{x.get('vgpr_spill_count')} -> {y.get('vgpr_count')}/{y.get('vgpr_spill_count')}{flag}")
print(f'{len(a)} kernels, {worse} with more spills in B')
