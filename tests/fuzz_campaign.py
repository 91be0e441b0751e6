"""tests/fuzz_campaign.py — the seeded fuzz tests of test_gpu_fuzz.py over many more seeds (not collected by pytest: run by hand on a GPU box,
`python3 tests/fuzz_campaign.py`; 40 seeds = 11600 cases take about a minute)."""
import sys
sys.path.insert(0, '.')
import dsc_amd
dsc_amd.init(12 << 30, 4 << 30)
import tests.test_gpu_fuzz as F
bad = 0
for seed in range(1000, 1040):
    for fn in (F.test_fuzz_transforms, F.test_fuzz_binary_ops, F.test_fuzz_reductions_and_unary):
        try:
            fn.__wrapped__(dsc_amd, seed) if hasattr(fn, '__wrapped__') else fn(dsc_amd, seed)
        except AssertionError as e:
            msg = str(e)
            if 'paths' in msg or 'done' in msg:      # the draw-coverage asserts of the test, not parity
                continue
            bad += 1
            print('FAIL', fn.__name__, seed, msg[:300], flush=True)
dsc_amd.synchronize()
print('FUZZ', 'FAILED' if bad else 'OK', bad)
