"""Timing probe: the 128 -> 64 layers of the 256^2 / 128^2 up path as (a) one generic launch on the concat view,
(b) two weight-stationary 64 -> 64 passes (second one takes the first one's output as its residual), and the folded
1x1 skip as a separate launch.  Run on the GPU box."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "diffusion-separation_amd"))
from diffsep_amd import ops

def tm(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

B = 16
for H in (256, 128):
    W = H
    g = torch.Generator(device="cuda").manual_seed(0)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    xa, xb = r(B, H, W, 64).bfloat16(), r(B, H, W, 64).bfloat16()
    w = r(64, 128, 3, 3) / (9 * 128) ** 0.5
    w1 = r(64, 128, 1, 1) / 128 ** 0.5
    bias, tb = r(64), r(B, 64)
    sc, sh = torch.rand(B, 128, device="cuda") + 0.5, r(B, 128) * 0.1
    res = r(B, H, W, 64).bfloat16()
    wp = ops.pack_conv_weight(w.cpu(), torch.bfloat16).cuda()
    wpa = ops.pack_conv_weight(w[:, :64].contiguous().cpu(), torch.bfloat16).cuda()
    wpb = ops.pack_conv_weight(w[:, 64:].contiguous().cpu(), torch.bfloat16).cuda()
    wp1 = ops.pack_conv_weight(w1.cpu(), torch.bfloat16).cuda()
    sca, sha, scb, shb = sc[:, :64].contiguous(), sh[:, :64].contiguous(), sc[:, 64:].contiguous(), sh[:, 64:].contiguous()
    st = torch.zeros(B, 64, 2, dtype=torch.int64, device="cuda")
    y = torch.zeros(B, H, W, 64, dtype=torch.bfloat16, device="cuda")
    part = torch.zeros_like(y)
    one = lambda: ops.conv2d_fused(xa, wp, bias, 64, 3, x2=xb, gn=(sc, sh), gn_act=1, bias_b=tb, out=y, stats=st)
    p1 = lambda: ops.conv2d_fused(xa, wpa, None, 64, 3, gn=(sca, sha), gn_act=1, out=part)
    p2 = lambda: ops.conv2d_fused(xb, wpb, bias, 64, 3, gn=(scb, shb), gn_act=1, bias_b=tb, res=part, out=y, stats=st)
    nin = lambda: ops.conv2d_fused(xa, wp1, bias, 64, 1, x2=xb, out=part)
    c2 = lambda: ops.conv2d_fused(xa, wpa, bias, 64, 3, gn=(sca, sha), gn_act=1, res=part, out_scale=0.7071, out=y, stats=st)
    y_one = one().float() if False else None
    one(); ya = y.clone().float()
    p1(); p2(); yb = y.clone().float()
    err = ((ya - yb).pow(2).mean() / ya.pow(2).mean()).sqrt().item()
    print(f"{H}^2 B={B}: generic 128->64 {tm(one):.1f} us | WS pass1 {tm(p1):.1f} + pass2 {tm(p2):.1f} us (rel rms between them {err:.2e})"
          f" | 1x1 128->64 {tm(nin):.1f} us, WS 64->64 + residual {tm(c2):.1f} us")
