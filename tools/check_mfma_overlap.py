#!/usr/bin/env python
"""Scan gfx950 assembly (hipcc -S --cuda-device-only, or llvm-objdump -d of an unbundled code object) for MFMA instructions whose destination registers overlap a source operand
without being identical to it.

hipcc 7.2 (ROCm 7.2.0) allocates v_mfma_f32_16x16x32_f16 with vDst PARTIALLY overlapping SrcC / SrcB (seen in csrc/chain.hip:
`v_mfma_f32_16x16x32_f16 v[220:223], v[90:93], v[218:221], v[222:225]`); on the MI355X that instruction produced wrong values in
the overlapping registers, nondeterministically (profiles/r04_mfma_overlap.txt).  The kernels that use the new MFMAs go through
wrappers that keep the result and all three sources live at one point, hence in disjoint registers (csrc/common.h ymi_mfma16 /
ymi_mfma32), and `make lint`
runs this scan over the disassembly of every built object so that a compiler or source change that brings the pattern back is
caught before it reaches the GPU.

    python tools/check_mfma_overlap.py file.s [file.s ...]      exit status 1 if any offending instruction is found
"""
import re
import sys


def rng(op):
    m = re.fullmatch(r'[va]\[(\d+):(\d+)\]', op)
    if m:
        return op[0], int(m.group(1)), int(m.group(2))
    m = re.fullmatch(r'[va](\d+)', op)
    if m:
        return op[0], int(m.group(1)), int(m.group(1))
    return None


def main():
    bad = 0
    for fn in sys.argv[1:]:
        kern = '?'
        for ln, line in enumerate(open(fn), 1):
            s = line.split('//')[0].strip()               # (llvm-objdump appends `// address: encoding`)
            if s.endswith(':') and not s.startswith('.') and not s.startswith(';'):
                kern = s[:-1].split('<')[-1].rstrip('>')
            if not s.startswith('v_mfma'):
                continue
            ops = [o.strip() for o in s.split(None, 1)[1].split(',')]
            d = rng(ops[0])
            if d is None:
                continue
            for k, o in enumerate(ops[1:4]):
                r = rng(o)
                if r is None or r[0] != d[0]:
                    continue
                overlap = not (r[2] < d[1] or r[1] > d[2])
                if overlap and not (k == 2 and r[1:] == d[1:]):
                    bad += 1
                    print('%s:%d [%s] %s   <- vDst overlaps src%s' % (fn, ln, kern[:60], s, 'ABC'[k]))
    print('%d offending MFMA instruction(s)' % bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
