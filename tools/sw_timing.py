#!/usr/bin/env python3
"""Per-phase tick totals (wave 0) of the streamed-weight kernel's profiling build (-DSW_TIMING; DIFFSEP_LIB_F16 points at it).
DIFFSEP_SW_DBG: bit 0 = the stores fall outside the tensor, bit 1 = the input loads do.  Usage: sw_timing.py [case filter]"""
import ctypes
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-separation_amd"))
from diffsep_amd import ops  # noqa: E402

l = ctypes.CDLL(os.environ["DIFFSEP_LIB_F16"])
names = ["prologue: staging of the first chunk", "barrier wait", "3x3 chunk phases", "skip chunk phases", "final epilogue", "tail (statistics)",
         "prologue: tables + descriptors", "prologue: first loads issued, tables, barrier"]
CASES = [("128->128 @256^2", 128, 0, None, 256), ("256->128 @256^2", 128, 128, None, 256), ("128->128 +skip256 @256^2", 128, 0, (128, 128), 256),
         ("256->128 @64^2", 128, 128, None, 64), ("128->128 @64^2", 128, 0, None, 64)]
DT, B, CO = torch.float16, 16, 128
for name, C1, C2, skip, H in CASES:
    if len(sys.argv) > 1 and sys.argv[1] not in name:
        continue
    W, C = H, C1 + C2
    a = torch.randn(B, H, W, C1, device="cuda").to(DT)
    bt = torch.randn(B, H, W, C2, device="cuda").to(DT) if C2 else None
    wf = ops.pack_frag_weight(torch.randn(CO, C, 3, 3) / (9 * C) ** 0.5, DT).cuda()
    bias = torch.randn(CO, device="cuda")
    sc, sh = torch.rand(B, C, device="cuda") + 0.5, torch.randn(B, C, device="cuda") * 0.1
    sk = None
    if skip:
        sk = (torch.randn(B, H, W, skip[0], device="cuda").to(DT), torch.randn(B, H, W, skip[1], device="cuda").to(DT),
              ops.pack_frag_weight(torch.randn(CO, sum(skip), 1, 1) / sum(skip) ** 0.5, DT).cuda())
    y = torch.zeros(B, H, W, CO, device="cuda", dtype=DT)
    st = torch.zeros((B, CO, 2), dtype=torch.int64, device="cuda")
    run = lambda: ops.conv3x3_streamed(a, wf, CO, x2=bt, gn=(sc, sh), bias=bias, skip=sk, stats=st, out=y)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    l.diffsep_sw_debug_read(out, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    l.diffsep_sw_debug_read(out, 1)
    nb = out[15]
    tot = sum(out[i] for i in range(8))
    us = e0.elapsed_time(e1) / 5 * 1e3
    print(f"{name}: {us:.1f} us/launch, {tot / nb:.0f} ticks/block (wave 0), {nb // 5} blocks -> {tot / nb / us / 1e3:.2f} ticks/ns")
    for i in range(8):
        print(f"    {names[i]:48s} {out[i] / nb:9.0f}  {100 * out[i] / tot:5.1f} %")
