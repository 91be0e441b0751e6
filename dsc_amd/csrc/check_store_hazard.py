#!/usr/bin/env python3
"""Build-time check for the 128-bit store-data hazard of gfx950 (fft_regs_common.h `buf_store`).

Observed on MI355X in round 2: `buffer_store_dwordx4 v[66:69], ...` (SGPR soffset) directly followed by a VALU instruction that
rewrites one of v[66:69] stored the NEW value in lanes 12-15 of every 16-lane row, once in a few thousand rows.  hipcc's hazard
recognizer pads that pattern only for a literal soffset (and with ONE wait state), so the library's c64 stores carry their own
two wait states — and this script proves, on the code objects that actually ship, that nothing slipped through:

    for every buffer_store_dwordx3 / x4 (and global / flat / scratch stores of the same width) no VALU instruction may write one
    of the store's data VGPRs before TWO wait states have passed (ONE for the flat forms, which hipcc pads itself);
    `s_nop N` counts N + 1, every other instruction 1.

Usage:  check_store_hazard.py <libdsc_mi355x.so | object file | code object> ...      (exit 1 and a listing if anything is found)
Used by the Makefile (after linking) and by tests/test_abi.py::test_no_store_data_hazard_in_the_shipped_code_objects.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
BUNDLE_MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'

_STORE = re.compile(r'^(buffer|global|flat|scratch)_store_(dwordx3|dwordx4|b96|b128)\s+(.*)$')
_VREG = re.compile(r'v\[(\d+):(\d+)\]|v(\d+)\b')
_NOP = re.compile(r'^s_nop\s+(\d+)')


def _vrange(tok):
    m = _VREG.match(tok.strip())
    if not m:
        return None
    if m.group(3) is not None:
        return int(m.group(3)), int(m.group(3))
    return int(m.group(1)), int(m.group(2))


def _valu_dest(instr):
    """VGPR range written by a VALU instruction, or None (compares into SGPRs / VCC, readlanes ... write no VGPR)."""
    if not instr.startswith('v_'):
        return None
    op, _, rest = instr.partition(' ')
    if op.startswith(('v_cmp', 'v_cmpx', 'v_readlane', 'v_readfirstlane', 'v_nop')):
        return None
    first = rest.split(',')[0]
    return _vrange(first)


def device_code_objects(path):
    """The gfx950 code objects inside a shared library / object file (clang offload bundles in .hip_fatbin), or the file itself
    when it already is an AMDGPU ELF.  Yields (label, bytes)."""
    data = open(path, 'rb').read()
    found = False
    pos = 0
    while True:
        i = data.find(BUNDLE_MAGIC, pos)
        if i < 0:
            break
        n = struct.unpack_from('<Q', data, i + 24)[0]
        off = i + 32
        for _ in range(n):
            o, size, tsz = struct.unpack_from('<QQQ', data, off)
            off += 24
            triple = data[off:off + tsz].decode()
            off += tsz
            if 'amdgcn' in triple and size:
                found = True
                yield f'{os.path.basename(path)}@{i + o}', data[i + o:i + o + size]
        pos = i + 24
    if not found and data[:4] == b'\x7fELF' and data[18:20] == b'\xe0\x00':        # e_machine = EM_AMDGPU
        yield os.path.basename(path), data


def disassemble(blob):
    with tempfile.NamedTemporaryFile(suffix='.co') as f:
        f.write(blob)
        f.flush()
        out = subprocess.run([f'{LLVM}/llvm-objdump', '-d', '--no-show-raw-insn', '--no-leading-addr', f.name], check=True, capture_output=True, text=True).stdout
    return out


def scan_text(asm, label=''):
    """Returns (number of wide stores, violations).  `asm` is llvm-objdump (or hipcc -S) text."""
    stores = 0
    bad = []
    func = '?'
    lines = []
    for raw in asm.splitlines():
        s = raw.split('//')[0].split(';')[0].strip()
        if not s:
            continue
        if s.endswith(':') and not s.startswith(('v_', 's_', 'buffer_', 'global_', 'ds_', 'flat_', 'scratch_')):
            func = s.rstrip(':').strip('<>')
            m = re.match(r'^[0-9a-f]+ <(.*)>$', func)
            if m:
                func = m.group(1)
            lines.append(('label', func))
            continue
        lines.append(('i', s))
    cur = '?'
    for idx, (kind, s) in enumerate(lines):
        if kind == 'label':
            if not s.startswith(('.L', 'L')):
                cur = s
            continue
        m = _STORE.match(s)
        if not m:
            continue
        ops = [o.strip() for o in m.group(3).split(',')]
        data = _vrange(ops[0] if m.group(1) == 'buffer' else (ops[1] if len(ops) > 1 else ''))
        if data is None:
            continue
        stores += 1
        need = 2 if m.group(1) == 'buffer' else 1
        ws = 0
        k = idx + 1
        while ws < need and k < len(lines):
            kind2, t = lines[k]
            k += 1
            if kind2 == 'label':
                continue
            d = _valu_dest(t)
            if d is not None and d[0] <= data[1] and d[1] >= data[0]:
                bad.append(f'{label} {cur}: `{s}` then, {ws} wait state(s) later, `{t}`')
                break
            nop = _NOP.match(t)
            ws += int(nop.group(1)) + 1 if nop else 1
            if t.startswith(('s_endpgm', 's_branch', 's_setpc')):
                break
    return stores, bad


def scan_file(path):
    total, bad = 0, []
    for label, blob in device_code_objects(path):
        n, b = scan_text(disassemble(blob), label)
        total += n
        bad += b
    return total, bad


def main(argv):
    total, bad = 0, []
    for p in argv:
        n, b = scan_file(p)
        total += n
        bad += b
    print(f'check_store_hazard: {total} stores of more than 64 bits, {len(bad)} with a VALU write to their data registers inside the wait states')
    for line in bad[:40]:
        print('  ' + line)
    return 1 if bad or total == 0 else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
