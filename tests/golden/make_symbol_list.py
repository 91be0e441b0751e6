#!/usr/bin/env python3
"""Regenerates tests/golden/reference_binding_symbols.txt: the NAMES the reference's Python bindings resolve in its shared
library (a fixture: names only, no reference source).  Run where /root/reference exists."""
import re
import sys

OUT_OF_SCOPE = {'dsc_arange', 'dsc_clip', 'dsc_concat', 'dsc_cos', 'dsc_exp', 'dsc_i0', 'dsc_log10', 'dsc_log2', 'dsc_logn', 'dsc_pow',
                'dsc_randn', 'dsc_reshape', 'dsc_sin', 'dsc_sinc', 'dsc_sqrt'}
src = open(sys.argv[1] if len(sys.argv) > 1 else '/root/reference/python/dsc/_bindings.py').read()
names = sorted(set(re.findall(r'_lib\.(dsc_[a-z0-9_]+)', src)))
for n in names:
    print(n, 'out' if n in OUT_OF_SCOPE else 'path')
