#!/usr/bin/env python3
"""Stand-alone timing of the thin convolutions (the output-pyramid heads Cin -> 6 + residual, the first layer 8 -> 64)
through the C-ABI, fp16 build, B = 16.  Usage: python tools/thin_bench.py [reps]"""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-separation_amd"))
from diffsep_amd import ops  # noqa: E402

DT = torch.float16
CASES = [(64, 6, 256), (64, 6, 128), (128, 6, 64), (128, 6, 256), (8, 64, 256)]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    B = 16
    for ci, co, H in CASES:
        W = H
        x = torch.randn(B, H, W, ci, device="cuda").to(DT)
        kc = 32 if ci >= 64 else 0
        w = (torch.randn(co, 9, ci, device="cuda") / (9 * ci) ** 0.5).to(DT)
        if kc:
            w = w.reshape(co, 9, ci // kc, kc).permute(2, 1, 0, 3).contiguous()
        b = torch.randn(co, device="cuda")
        cp = (co + 7) // 8 * 8
        y = torch.zeros(B, H, W, cp, device="cuda", dtype=DT)
        if ci >= 64:
            sc, sh = torch.rand(B, ci, device="cuda") + 0.5, torch.randn(B, ci, device="cuda") * 0.1
            res = torch.randn(B, H, W, cp, device="cuda").to(DT)
            run = lambda: ops.conv2d_fused(x, w, b, co, 3, gn=(sc, sh), gn_act=1, res=res, cout_pad=cp, out=y, w_chunk=kc)
        else:
            st = torch.zeros((B, co, 2), dtype=torch.int64, device="cuda")
            run = lambda: ops.conv2d_fused(x, w, b, co, 3, cout_pad=cp, out=y, stats=st)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        byt = B * H * W * (ci + 2 * cp if ci >= 64 else ci + co) * 2
        print(f"3x3 {ci:4d}->{co:3d} {H:4d}^2: {us:8.1f} us  {byt / us / 1e6:6.2f} TB/s algorithmic")


if __name__ == "__main__":
    main()
