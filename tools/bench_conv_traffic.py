#!/usr/bin/env python3
"""How the 64->64 3x3 conv launch time moves with its HBM traffic (residual on/off, statistics on/off, batch)."""
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

dt = torch.bfloat16
for (B, H, W) in [(16, 256, 256), (8, 256, 256), (32, 256, 256), (16, 128, 128)]:
    x = torch.randn(B, H, W, 64, device="cuda").to(dt)
    w = (torch.randn(64, 9, 64, device="cuda") / 24).to(dt)
    b = torch.randn(64, device="cuda")
    sc, sh = torch.rand(B, 64, device="cuda") + 0.5, torch.randn(B, 64, device="cuda") * 0.1
    res = torch.randn(B, H, W, 64, device="cuda").to(dt)
    y = torch.zeros(B, H, W, 64, device="cuda", dtype=dt)
    _, st = ops.conv2d_fused(x, w, b, 64, 3, out=y, stats=True)
    mb = x.numel() * 2 / 1e6
    for name, kw, traffic in [("gn+res+stats", dict(gn=(sc, sh), gn_act=1, res=res, stats=st), 3),
                              ("gn+stats", dict(gn=(sc, sh), gn_act=1, stats=st), 2),
                              ("gn", dict(gn=(sc, sh), gn_act=1), 2), ("plain", dict(), 2)]:
        us = t(lambda: ops.conv2d_fused(x, w, b, 64, 3, out=y, out_scale=0.7071, **kw))
        print(f"B={B:2d} {H}x{W} {name:14s}: {us:7.1f} us  {traffic*mb/us/1e6*1e6/1e6:5.2f} TB/s algorithmic")
