#!/usr/bin/env python3
"""A/B of the split mode's streamed-weight kernel (conv3x3_sws.hip) against the generic tile in split mode (conv_mfma.hip, SP = 1)
through the C-ABI, fp32 tensors, B = 16.  Usage: python tools/sws_bench.py [reps] [substring of the case names]"""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-separation_amd"))
from diffsep_amd import ops  # noqa: E402

CASES = [  # (name, C1, C2, Cout, H, W, raw, res)
    ("64->64 @256^2", 64, 0, 64, 256, 256, False, False), ("64->64 +res @256^2", 64, 0, 64, 256, 256, False, True),
    ("128->64 @256^2", 64, 64, 64, 256, 256, False, False), ("64->64 raw @256^2", 64, 0, 64, 256, 256, True, False),
    ("64->64 @128^2", 64, 0, 64, 128, 128, False, False), ("128->64 @128^2", 64, 64, 64, 128, 128, False, False),
    ("128->128 @64^2", 128, 0, 128, 64, 64, False, False), ("256->128 @64^2", 128, 128, 128, 64, 64, False, False),
    ("128->128 +res @64^2", 128, 0, 128, 64, 64, False, True),
    ("128->128 @256^2", 128, 0, 128, 256, 256, False, False), ("256->128 @256^2", 128, 128, 128, 256, 256, False, False),
]


def timeit(run, reps):
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    B = 16
    for name, C1, C2, CO, H, W, raw, res in CASES:
        if len(sys.argv) > 2 and sys.argv[2] not in name:
            continue
        C = C1 + C2
        a = torch.randn(B, H, W, C1, device="cuda")
        bt = torch.randn(B, H, W, C2, device="cuda") if C2 else None
        w4 = torch.randn(CO, C, 3, 3) / (9 * C) ** 0.5
        wf = ops.pack_frag_weight_split(w4).cuda()
        kc = ops.conv2d_chunk(3, torch.float32)
        wk = ops.pack_conv_weight(w4, torch.float32, chunk=kc).cuda()
        bias, bb = torch.randn(CO, device="cuda"), torch.randn(B, CO, device="cuda")
        sc, sh = torch.rand(B, C, device="cuda") + 0.5, torch.randn(B, C, device="cuda") * 0.1
        gn = None if raw else (sc, sh)
        r = torch.randn(B, H, W, CO, device="cuda") if res else None
        ident = ops.pack_frag_weight_split(torch.eye(CO).reshape(CO, CO, 1, 1)).cuda() if res else None
        y = torch.zeros(B, H, W, CO, device="cuda")
        st = torch.zeros((B, CO, 2), dtype=torch.int64, device="cuda")
        fl = 2.0 * 9 * C * CO * H * W * B
        us_new = timeit(lambda: ops.conv3x3_streamed(a, wf, CO, x2=bt, gn=gn, bias=bias, bias_b=bb, stats=st, out=y, res=r, ident_frag=ident), reps)
        us_old = timeit(lambda: ops.conv2d_fused(a, wk, bias, CO, 3, x2=bt, gn=gn, gn_act=0 if raw else 1, bias_b=bb, res=r, out=y,
                                                 stats=st, w_chunk=kc, split=True), reps)
        print(f"{name:24s} streamed split {us_new:7.1f} us {3 * fl / us_new / 1e6:7.1f} bf16 TF/s ({3 * fl / us_new / 1e6 / 2500:.3f}) | generic tile "
              f"{us_old:7.1f} us ({3 * fl / us_old / 1e6 / 2500:.3f})", flush=True)


if __name__ == "__main__":
    main()
