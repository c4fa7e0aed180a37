#!/usr/bin/env python3
"""Count, per kernel of a `hipcc -S --cuda-device-only` file, the sites where a scalar VALU instruction writes a VGPR
that the very next instruction reads as half of a packed-FP32 operand (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32,
optionally only with op_sel).  This is the shape of the GroupNorm-statistics code that returned a wrong sum lane
under co-resident kernels (profiles/experiments/README.md, "Streams"); after the fix the conv kernels have none
with op_sel.  usage: isa_scan_pk.py file.s [--any]"""
import re
import sys

PK = re.compile(r'^\s*(v_pk_fma_f32|v_pk_mul_f32|v_pk_add_f32)\s+v\[(\d+):(\d+)\],\s*(.*)$')


def regs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok.strip().split(' ')[0])
    return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()


def main():
    need_opsel = "--any" not in sys.argv
    kern, cnt, prev = None, {}, None
    for l in open(sys.argv[1]):
        if re.match(r'^_Z\w+:', l):
            kern = l.split(':')[0]
        t = l.strip()
        if not t or t.startswith((';', '.')):
            continue
        m = PK.match(l)
        if m and prev:
            srcs = set().union(*[regs(o) for o in m.group(4).split(',')[:3]])
            pm = re.match(r'^\s*(v_(?!pk_)\w+)\s+v(\d+),', prev)
            if pm and int(pm.group(2)) in srcs and ('op_sel' in l or not need_opsel):
                cnt[kern] = cnt.get(kern, 0) + 1
        prev = l
    for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
        print(f"{v:5d}  {k}")
    print(f"{sum(cnt.values())} sites in {len(cnt)} kernels")


if __name__ == "__main__":
    main()
