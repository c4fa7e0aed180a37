#!/usr/bin/env python3
"""Time the fused GroupNorm + SiLU + FIR x2 up / down kernels at the sizes of the two largest ResBlocks."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-separation_amd"))
from diffsep_amd import ops
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (C, H, W, mode) in [(64, 128, 128, 1), (64, 256, 256, 2), (128, 64, 64, 1), (128, 128, 128, 2), (64, 128, 128, 2), (128, 64, 64, 2)]:
    x = torch.randn(16, H, W, C, device="cuda").to(torch.float16)
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    us = t(lambda: ops.groupnorm_act(x, g, b, min(C // 4, 32), 1e-6, 1, mode, want_xr=True))
    print(f"C={C} {H}x{W} mode={mode}: {us:.1f} us (incl. statistics passes)")
