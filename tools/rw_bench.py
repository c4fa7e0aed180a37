#!/usr/bin/env python3
"""A/B of the register-weight 3x3 kernel (conv3x3_rw.hip) against the kernels it replaces, through the C-ABI, on the
launch shapes of the 256^2 / 128^2 levels (nf = 64, B = 16).  DIFFSEP_NO_RW=1 selects the old kernels.
Usage: python tools/rw_bench.py [reps] [substring of the case names to run]"""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-separation_amd"))
from diffsep_amd import ops  # noqa: E402

DT = torch.float16 if os.environ.get("RW_DT") == "f16" else torch.bfloat16  # (RW_DT=f16: the half-precision build)
CASES = [  # (name, C1, C2, H, W, mode)   mode: conv0 = GN+SiLU, bias, temb, stats; conv1res = + residual, 1/sqrt(2); plain
    ("64->64 conv0", 64, 0, 256, 256, "conv0"), ("64->64 conv1+res", 64, 0, 256, 256, "res"), ("64->64 plain", 64, 0, 256, 256, "plain"),
    ("cat(64,64)->64 conv0", 64, 64, 256, 256, "conv0"), ("cat(64,64)->64 plain", 64, 64, 256, 256, "plain"),
    ("64->64 conv0 128^2", 64, 0, 128, 128, "conv0"), ("cat(64,64)->64 conv0 128^2", 64, 64, 128, 128, "conv0"),
    # 128 couts (DIFFSEP_NO_RW128=1 selects the generic tile): nf = 128 at 256^2, nf = 64 at 128^2 / 64^2 / 32^2
    ("128->128 conv0 256^2", 128, 0, 256, 256, "conv0", 128), ("128->128 conv1+res 256^2", 128, 0, 256, 256, "res", 128),
    ("128->128 plain 256^2", 128, 0, 256, 256, "plain", 128), ("128->128 conv0 128^2", 128, 0, 128, 128, "conv0", 128),
    ("128->128 conv0 64^2", 128, 0, 64, 64, "conv0", 128), ("128->128 conv1+res 64^2", 128, 0, 64, 64, "res", 128),
    ("128->128 conv0 32^2", 128, 0, 32, 32, "conv0", 128),
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    B = 16
    for name, C1, C2, H, W, mode, *rest in CASES:
        C, CO = C1 + C2, (rest[0] if rest else 64)
        if len(sys.argv) > 2 and sys.argv[2] not in name:
            continue
        a = torch.randn(B, H, W, C1, device="cuda").to(DT)
        bt = torch.randn(B, H, W, C2, device="cuda").to(DT) if C2 else None
        kc = ops.conv2d_chunk(3, DT)
        w = (torch.randn(CO, 9, C, device="cuda") / (9 * C) ** 0.5).to(DT)
        w = w.reshape(CO, 9, C // kc, kc).permute(2, 1, 0, 3).contiguous()
        bias, bb = torch.randn(CO, device="cuda"), torch.randn(B, CO, device="cuda")
        sc, sh = torch.rand(B, C, device="cuda") + 0.5, torch.randn(B, C, device="cuda") * 0.1
        res = torch.randn(B, H, W, CO, device="cuda").to(DT)
        y = torch.zeros(B, H, W, CO, device="cuda", dtype=DT)
        st = torch.zeros((B, CO, 2), dtype=torch.int64, device="cuda")
        if mode == "conv0":
            run = lambda: ops.conv2d_fused(a, w, bias, CO, 3, x2=bt, gn=(sc, sh), gn_act=1, bias_b=bb, out=y, stats=st, w_chunk=kc)
        elif mode == "res":
            run = lambda: ops.conv2d_fused(a, w, bias, CO, 3, x2=bt, gn=(sc, sh), gn_act=1, bias_b=bb, res=res, out_scale=0.7071, out=y, stats=st, w_chunk=kc)
        else:
            run = lambda: ops.conv2d_fused(a, w, None, CO, 3, x2=bt, out=y, w_chunk=kc)
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
        fl = 2.0 * 9 * C * CO * H * W * B
        print(f"{name:30s} {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s  ({fl / us / 1e6 / 2500:.3f} of the MFMA peak)")


if __name__ == "__main__":
    main()
