#!/usr/bin/env python3
"""A/B of the streamed-weight 3x3 kernel (conv3x3_sw.hip, unit entry diffsep_conv3x3_streamed) against what the dispatch runs
for the same launch without it (DIFFSEP_NO_SW: generic tile / register-weight kernel), through the C-ABI, B = 16, fp16 build.
The folded skip has no stand-alone counterpart at the unit entry points: its rows time the streamed kernel only.
Usage: python tools/sw_bench.py [reps] [substring of the case names]"""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-separation_amd"))
from diffsep_amd import _lib, ops  # noqa: E402

DT = torch.float16
CASES = [  # (name, C1, C2, skip (s1, s2) | None, H, W, raw input)
    ("256->128 @64^2", 128, 128, None, 64, 64, False), ("192->128 @64^2", 128, 64, None, 64, 64, False),
    ("128->128 @64^2", 128, 0, None, 64, 64, False), ("64->128 raw @64^2", 64, 0, None, 64, 64, True),
    ("128->128 +skip256 @64^2", 128, 0, (128, 128), 64, 64, False), ("128->128 +skip192 @64^2", 128, 0, (128, 64), 64, 64, False),
    ("128->128 +skip128 @64^2", 128, 0, (128, 0), 64, 64, False), ("128->128 +skip64 @64^2", 128, 0, (64, 0), 64, 64, False),
    ("256->128 @128^2", 128, 128, None, 128, 128, False), ("128->128 +skip256 @128^2", 128, 0, (128, 128), 128, 128, False),
    ("128->128 @128^2", 128, 0, None, 128, 128, False),
    ("256->128 @256^2", 128, 128, None, 256, 256, False), ("128->128 +skip256 @256^2", 128, 0, (128, 128), 256, 256, False),
    ("128->128 @256^2", 128, 0, None, 256, 256, False),
    ("256->128 @32^2", 128, 128, None, 32, 32, False), ("128->128 @32^2", 128, 0, None, 32, 32, False),
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
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    B, CO = 16, 128
    L = _lib.lib("f16")
    for name, C1, C2, skip, H, W, raw in CASES:
        if len(sys.argv) > 2 and sys.argv[2] not in name:
            continue
        C = C1 + C2
        a = torch.randn(B, H, W, C1, device="cuda").to(DT)
        bt = torch.randn(B, H, W, C2, device="cuda").to(DT) if C2 else None
        w4 = torch.randn(CO, C, 3, 3) / (9 * C) ** 0.5
        wf = ops.pack_frag_weight(w4, DT).cuda()
        kc = ops.conv2d_chunk(3, DT)
        wk = ops.pack_conv_weight(w4, DT, chunk=kc).cuda()
        bias, bb = torch.randn(CO, device="cuda"), torch.randn(B, CO, device="cuda")
        sc, sh = torch.rand(B, C, device="cuda") + 0.5, torch.randn(B, C, device="cuda") * 0.1
        gn = None if raw else (sc, sh)
        sk = None
        fl = 2.0 * 9 * C * CO * H * W * B
        if skip:
            s1, s2 = skip
            sa = torch.randn(B, H, W, s1, device="cuda").to(DT)
            sb = torch.randn(B, H, W, s2, device="cuda").to(DT) if s2 else None
            sk = (sa, sb, ops.pack_frag_weight(torch.randn(CO, s1 + s2, 1, 1) / (s1 + s2) ** 0.5, DT).cuda())
            fl += 2.0 * (s1 + s2) * CO * H * W * B
        y = torch.zeros(B, H, W, CO, device="cuda", dtype=DT)
        st = torch.zeros((B, CO, 2), dtype=torch.int64, device="cuda")
        us_sw = timeit(lambda: ops.conv3x3_streamed(a, wf, CO, x2=bt, gn=gn, bias=bias, bias_b=bb, skip=sk, stats=st, out=y), reps)
        line = f"{name:28s} streamed {us_sw:7.1f} us {fl / us_sw / 1e6:7.1f} TF/s ({fl / us_sw / 1e6 / 2500:.3f})"
        if not skip:
            _lib.check(L.diffsep_set_option(b"no_sw", 1), L)
            us_old = timeit(lambda: ops.conv2d_fused(a, wk, bias, CO, 3, x2=bt, gn=gn, gn_act=0 if raw else 1, bias_b=bb, out=y,
                                                     stats=st, w_chunk=kc), reps)
            _lib.check(L.diffsep_set_option(b"no_sw", 0), L)
            line += f" | dispatch without it {us_old:7.1f} us ({fl / us_old / 1e6 / 2500:.3f})"
        print(line, flush=True)


if __name__ == "__main__":
    main()
