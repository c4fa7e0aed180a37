#!/usr/bin/env python3
"""Micro-benchmark of the MFMA conv kernel through the C-ABI on the layer shapes of NCSN++ (nf=64, B=16, 4 s).
Usage: python tools/bench_conv.py [bf16|f32] [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-separation_amd"))
from diffsep_amd import ops  # noqa: E402

SHAPES = [  # (ksize, Cin, Cout, H, W, count per forward)
    (3, 64, 64, 256, 256, 9), (3, 128, 64, 256, 256, 3), (3, 64, 64, 128, 128, 9), (3, 128, 128, 128, 128, 2),
    (3, 192, 64, 128, 128, 1), (3, 128, 128, 64, 64, 8), (3, 256, 128, 64, 64, 2), (3, 128, 128, 32, 32, 8),
    (3, 256, 128, 32, 32, 3), (3, 128, 128, 16, 16, 8), (3, 256, 128, 16, 16, 3), (3, 128, 128, 4, 4, 12),
    (3, 64, 6, 256, 256, 1), (3, 8, 64, 256, 256, 1), (1, 128, 64, 256, 256, 3), (1, 64, 64, 128, 128, 2),
    (1, 128, 128, 16, 16, 16),
]


if os.environ.get("BENCH_CONV_SHAPES"):  # "k,Cin,Cout,H,W;..." replaces the table
    SHAPES = [tuple(int(v) for v in t.split(",")) + (1,) for t in os.environ["BENCH_CONV_SHAPES"].split(";")]
FUSED = os.environ.get("BENCH_CONV_PLAIN", "0") != "1"
CHUNKED = os.environ.get("BENCH_CONV_CHUNKED", "0") == "1"


def main():
    dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    sel = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else None
    B = 16
    tot_t = tot_f = 0.0
    for idx, (k, ci, co, H, W, cnt) in enumerate(SHAPES):
        if sel is not None and idx not in sel:
            continue
        x = torch.randn(B, H, W, ci, device="cuda").to(dt)
        w = (torch.randn(co, k * k, ci, device="cuda") / (k * k * ci) ** 0.5).to(dt)
        kc = ops.conv2d_chunk(k, dt) if (CHUNKED and ci % ops.conv2d_chunk(k, dt) == 0) else 0
        if kc:
            w = w.reshape(co, k * k, ci // kc, kc).permute(2, 1, 0, 3).contiguous()
        b = torch.randn(co, device="cuda")
        cp = (co + 7) // 8 * 8
        y = torch.zeros(B, H, W, cp, device="cuda", dtype=dt)
        if FUSED and ci >= 64:  # GroupNorm + SiLU on the input, residual add: the shape of the engine's launches
            sc, sh = torch.rand(B, ci, device="cuda") + 0.5, torch.randn(B, ci, device="cuda") * 0.1
            res = torch.randn(B, H, W, cp, device="cuda").to(dt)
            _, st = ops.conv2d_fused(x, w, b, co, k, cout_pad=cp, out=y, stats=True, w_chunk=kc)  # int64 accumulators (keep adding: timing only)
            run = lambda: ops.conv2d_fused(x, w, b, co, k, gn=(sc, sh), gn_act=1, res=res, out_scale=0.7071, cout_pad=cp, out=y, stats=st, w_chunk=kc)
        else:
            run = lambda: ops.conv2d_fused(x, w, b, co, k, cout_pad=cp, out=y, w_chunk=kc)
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
        fl = 2.0 * k * k * ci * co * H * W * B
        print(f"k{k} {ci:4d}->{co:4d} {H:4d}x{W:<4d} x{cnt:2d}: {us:9.1f} us  {fl/us/1e6:8.1f} TF/s")
        tot_t += us * cnt
        tot_f += fl * cnt
    print(f"weighted total {tot_t/1e3:.2f} ms per forward, {tot_f/tot_t/1e6:.1f} TF/s")


if __name__ == "__main__":
    main()
