"""GPU parity of the register-weight 3x3 kernel (conv3x3_rw.hip: 64 / cat(64, 64) -> 64 couts at >= 32-row images, bf16)
through the C-ABI against torch fp32 on the CPU (same bf16-rounded operands) and, for whole residual blocks with the
folded 1x1 skip, against the CPU oracle.  Tolerance: 4e-3 relative RMS per convolution (bf16 storage of the activated
input and of the output), 1.5e-2 per residual block."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import diffsep_oracle as O
from diffsep_amd import ops, synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(autouse=True, scope="module")
def _every_eligible_launch_on_the_rw_kernel():
    """The dispatch keeps small images (64-channel launches at <= 128^2, 128-cout launches with fewer tiles than CUs) on
    the weight-stationary / generic kernels (faster there); this module tests the register-weight kernel on them too."""
    from diffsep_amd import _lib
    libs = [_lib.lib(k) for k in ("bf16", "f16")]
    for l in libs:
        _lib.check(l.diffsep_set_option(b"rw_small", 1), l)
    yield
    for l in libs:
        _lib.check(l.diffsep_set_option(b"rw_small", 0), l)
DEV = "cuda"
DT = torch.bfloat16


def rel_rms(a, b):
    a = a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = b.detach().double().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / (np.sqrt(np.mean(b ** 2)) + 1e-30))


def rnd(tag, shape, scale=1.0):
    return torch.from_numpy(synth.synth_noise(tag, shape)) * scale


# (B, H, W): one tile per block, several tiles per block (B >= 64 on 256 CUs), 8-row tiles (H % 16 != 0), wide images
SHAPES = [(2, 32, 32), (3, 64, 96), (64, 64, 64), (256, 32, 64), (2, 40, 64), (130, 24 + 16, 32), (1, 128, 256)]


@pytest.mark.parametrize("B,H,W", SHAPES)
@pytest.mark.parametrize("C1,C2", [(64, 0), (64, 64), (128, 0)])
@pytest.mark.parametrize("act", [1, None])
def test_rw_conv3x3_matches_torch(B, H, W, C1, C2, act):
    if B * H * W * (C1 + C2) > 40e6 and act is None:
        pytest.skip("large case covered with the activation")
    C = C1 + C2
    a = (rnd(f"rw.a{B}{H}{C1}", (B, H, W, C1), 1.2) + 0.1).to(DEV, DT)
    bt = (rnd(f"rw.b{B}{H}{C2}", (B, H, W, C2), 0.9) - 0.2).to(DEV, DT) if C2 else None
    w = rnd(f"rw.w{C}", (64, C, 3, 3), 1.0 / math.sqrt(9 * C))
    bias, bb = rnd("rw.bias", (64,), 0.1).to(DEV), rnd(f"rw.bb{B}", (B, 64), 0.1).to(DEV)
    # (a residual rides through the kernel as identity-weight skip chunks; no 128-channel layer of the network has one)
    res = rnd(f"rw.r{B}{H}", (B, H, W, 64)).to(DEV, DT) if C == 64 else None
    sc = (1.0 + rnd(f"rw.sc{B}{C}", (B, C), 0.2)).to(DEV)
    sh = rnd(f"rw.sh{B}{C}", (B, C), 0.2).to(DEV)
    xf = torch.cat([a.float(), bt.float()], -1) if C2 else a.float()
    if act is not None:
        xf = F.silu(xf * sc[:, None, None, :] + sh[:, None, None, :]).to(DT).float()
    wq = w.to(DT).float()
    ref = F.conv2d(xf.cpu().permute(0, 3, 1, 2), wq, bias.cpu(), padding=1).permute(0, 2, 3, 1)
    ref = (ref + bb.cpu()[:, None, None, :] + (res.float().cpu() if res is not None else 0.0)) * 0.70710678
    for chunk in (0, ops.conv2d_chunk(3, DT)):
        wp = ops.pack_conv_weight(w, DT, chunk=chunk).to(DEV) if chunk else ops.pack_conv_weight(w, DT).to(DEV)
        y, st = ops.conv2d_fused(a, wp, bias, 64, 3, x2=bt, gn=None if act is None else (sc, sh), gn_act=act or 0,
                                 bias_b=bb, res=res, out_scale=0.70710678, stats=True, w_chunk=chunk)
        assert rel_rms(y.float(), ref) < 4e-3
        s = ops.stats_to_float(st)
        assert torch.allclose(s[..., 0].cpu(), ref.double().sum((1, 2)), rtol=2e-3, atol=2e-3 * H * W)
        assert torch.allclose(s[..., 1].cpu(), (ref.double() ** 2).sum((1, 2)), rtol=2e-3, atol=2e-3 * H * W)
    # plain launch: no bias / residual / statistics / GroupNorm
    y2 = ops.conv2d_fused(a, ops.pack_conv_weight(w, DT).to(DEV), None, 64, 3, x2=bt)
    xr = torch.cat([a.float(), bt.float()], -1) if C2 else a.float()
    ref2 = F.conv2d(xr.cpu().permute(0, 3, 1, 2), wq, None, padding=1).permute(0, 2, 3, 1)
    assert rel_rms(y2.float(), ref2) < 4e-3


# ---- 128 -> 128 couts: 4 cout groups on one 4 x 32 pixel group (tiles of 4 rows), skip / residual fragments in LDS
@pytest.mark.parametrize("B,H,W", [(2, 32, 32), (3, 64, 96), (16, 64, 64), (2, 40, 64), (1, 128, 256), (130, 32, 32)])
@pytest.mark.parametrize("C1,C2", [(128, 0), (64, 64)])
@pytest.mark.parametrize("extra", ["none", "res"])  # (the folded skips: test_rw128_resblock_vs_oracle)
def test_rw128_conv3x3_matches_torch(B, H, W, C1, C2, extra):
    C, CO = C1 + C2, 128
    if extra != "none" and (B, H, W) in ((1, 128, 256), (130, 32, 32)) and C2:
        pytest.skip("covered by the single-tensor input")
    a = (rnd(f"rw8.a{B}{H}{C1}", (B, H, W, C1), 1.2) + 0.1).to(DEV, DT)
    bt = (rnd(f"rw8.b{B}{H}{C2}", (B, H, W, C2), 0.9) - 0.2).to(DEV, DT) if C2 else None
    w = rnd(f"rw8.w{C}", (CO, C, 3, 3), 1.0 / math.sqrt(9 * C))
    bias, bb = rnd("rw8.bias", (CO,), 0.1).to(DEV), rnd(f"rw8.bb{B}", (B, CO), 0.1).to(DEV)
    sc = (1.0 + rnd(f"rw8.sc{B}{C}", (B, C), 0.2)).to(DEV)
    sh = rnd(f"rw8.sh{B}{C}", (B, C), 0.2).to(DEV)
    xf = torch.cat([a.float(), bt.float()], -1) if C2 else a.float()
    xa = F.silu(xf * sc[:, None, None, :] + sh[:, None, None, :]).to(DT).float()
    wq = w.to(DT).float()
    ref = F.conv2d(xa.cpu().permute(0, 3, 1, 2), wq, bias.cpu(), padding=1).permute(0, 2, 3, 1) + bb.cpu()[:, None, None, :]
    kw = {}
    if extra == "res":
        res = rnd(f"rw8.r{B}{H}", (B, H, W, CO)).to(DEV, DT)
        ref = ref + res.float().cpu()
        kw = dict(res=res)
    ref = ref * 0.70710678
    for chunk in (0, ops.conv2d_chunk(3, DT)):
        wp = ops.pack_conv_weight(w, DT, chunk=chunk).to(DEV) if chunk else ops.pack_conv_weight(w, DT).to(DEV)
        y, st = ops.conv2d_fused(a, wp, bias, CO, 3, x2=bt, gn=(sc, sh), gn_act=1, bias_b=bb, out_scale=0.70710678, stats=True,
                                 w_chunk=chunk, **kw)
        assert rel_rms(y.float(), ref) < 4e-3
        s = ops.stats_to_float(st)
        assert torch.allclose(s[..., 0].cpu(), ref.double().sum((1, 2)), rtol=2e-3, atol=2e-3 * H * W)
        assert torch.allclose(s[..., 1].cpu(), (ref.double() ** 2).sum((1, 2)), rtol=2e-3, atol=2e-3 * H * W)
    if extra == "none":  # raw input (the launch behind a FIR resampling kernel): no GroupNorm, no bias, no statistics
        y2 = ops.conv2d_fused(a, ops.pack_conv_weight(w, DT).to(DEV), None, CO, 3, x2=bt)
        ref2 = F.conv2d(xf.cpu().permute(0, 3, 1, 2), wq, None, padding=1).permute(0, 2, 3, 1)
        assert rel_rms(y2.float(), ref2) < 4e-3


def test_rw_conv3x3_zero_padding_is_exact():
    # an all-ones image through GroupNorm + SiLU with shift: every border pixel must see ZERO padding (not silu(shift))
    B, H, W, C = 2, 32, 64, 64
    x = torch.ones((B, H, W, C)).to(DEV, DT)
    w = torch.ones((64, C, 3, 3)) / 64.0
    sc, sh = torch.ones((B, C)).to(DEV), torch.full((B, C), 0.5).to(DEV)
    y = ops.conv2d_fused(x, ops.pack_conv_weight(w, DT).to(DEV), None, 64, 3, gn=(sc, sh), gn_act=1)
    v = float(F.silu(torch.tensor(1.5)).to(DT))
    cnt = F.conv2d(torch.ones(1, 1, H, W), torch.ones(1, 1, 3, 3), padding=1)[0, 0]  # taps inside the image: 4 / 6 / 9
    ref = (cnt * v)[None, :, :, None].expand(B, H, W, 64)
    assert rel_rms(y.float(), ref) < 4e-3  # (one bf16 rounding of the output)
    assert float((y.float().cpu() - ref).abs().max()) < 0.05


@pytest.mark.parametrize("C1,C2", [(64, 0), (64, 64)])
def test_rw_conv3x3_groupnorm_from_producer_accumulators(C1, C2):
    B, H, W = 3, 32, 64
    C = C1 + C2
    groups = min(C // 4, 32)
    g, be = (1.0 + rnd(f"rwa.g{C}", (C,), 0.2)).to(DEV), rnd(f"rwa.be{C}", (C,), 0.1).to(DEV)

    def produce(tag, Cp):
        xi = rnd(f"rwa.x{tag}{Cp}", (B, H, W, 16)).to(DEV, DT)
        wi = ops.pack_conv_weight(rnd(f"rwa.w{tag}{Cp}", (Cp, 16, 3, 3), 1.0 / 12.0), DT).to(DEV)
        return ops.conv2d_fused(xi, wi, rnd(f"rwa.b{tag}{Cp}", (Cp,), 0.3).to(DEV), Cp, 3, stats=True)

    a, sa = produce("a", C1)
    (bt, sb) = produce("b", C2) if C2 else (None, None)
    w = rnd(f"rwa.w{C}", (64, C, 3, 3), 1.0 / math.sqrt(9 * C))
    wp = ops.pack_conv_weight(w, DT).to(DEV)
    y = ops.conv2d_fused(a, wp, None, 64, 3, x2=bt, gn_acc=(sa, sb, g, be, groups), gn_act=1)
    # reference on the CPU in fp32 (torch CPU ops, not MIOpen on the GPU), on the same 16-bit-rounded operands
    xcat = (torch.cat([a.float(), bt.float()], -1) if C2 else a.float()).cpu()
    hn = F.silu(F.group_norm(xcat.permute(0, 3, 1, 2), groups, g.cpu(), be.cpu(), eps=1e-6)).to(DT).float()
    ref = F.conv2d(hn, w.to(DT).float(), None, padding=1).permute(0, 2, 3, 1)
    assert rel_rms(y.float(), ref) < 1e-2


RB = [("GroupNorm_0.weight", "cin"), ("GroupNorm_0.bias", "cin"), ("Conv_0.weight", "w0"), ("Conv_0.bias", "cout"),
      ("Dense_0.weight", "d"), ("Dense_0.bias", "cout"), ("GroupNorm_1.weight", "cout"), ("GroupNorm_1.bias", "cout"),
      ("Conv_1.weight", "w1"), ("Conv_1.bias", "cout"), ("Conv_2.weight", "w2"), ("Conv_2.bias", "cout")]


@pytest.mark.parametrize("cin,B,H,W,up,down", [(128, 3, 32, 64, False, False), (64, 2, 64, 32, False, False),
                                               (128, 2, 32, 32, True, False), (128, 2, 64, 128, False, True), (256, 2, 32, 64, False, False)])
def test_rw128_resblock_vs_oracle(cin, B, H, W, up, down):
    # 128-cout residual blocks: plain (Conv_1 + residual), widening 64 -> 128 (Conv_1 + folded 64-channel skip), FIR up /
    # down (raw-input Conv_0, Conv_1 + folded 128-channel skip), 256 -> 128 (Conv_0 and the 256-channel skip stay on the
    # generic tile)
    cout = 128
    shp = dict(cin=(cin,), cout=(cout,), w0=(cout, cin, 3, 3), w1=(cout, cout, 3, 3), w2=(cout, cin, 1, 1), d=(cout, 64))
    tbl = [(n, shp[k]) for n, k in RB if not (n.startswith("Conv_2") and cin == cout and not up and not down)]
    sd = synth.synth_state_dict(tbl, 33)
    if "Conv_2.weight" in sd:
        sd["Conv_2.weight"] = (sd["Conv_2.weight"] * 3.0).astype(np.float32)
    x, temb = rnd(f"rw8rb.x{cin}{H}{W}{B}", (B, cin, H, W)), rnd("rw8rb.t", (B, 64))
    ref = O._res_block(O.to_torch(sd), "", x, temb, up=up, down=down)
    yb = ops.resblock_forward([sd[n] for n, _ in tbl], ops.to_nhwc(x).to(DT).to(DEV), temb.to(DEV), cout, up=up, down=down)
    assert rel_rms(ops.to_nchw(yb).float(), ref) < 1.5e-2


@pytest.mark.parametrize("cin,B,H,W,up,down", [(128, 3, 32, 64, False, False), (128, 2, 40, 32, False, False),
                                               (64, 3, 32, 32, True, False), (64, 2, 64, 128, False, True),
                                               (128, 70, 64, 64, False, False), (64, 2, 24, 32, True, False),
                                               (192, 2, 32, 64, False, False), (192, 33, 32, 32, False, False)])
def test_rw_resblock_with_folded_skip_vs_oracle(cin, B, H, W, up, down):
    # 64-cout residual blocks whose Conv_0 (cat(64, 64) -> 64, GroupNorm + SiLU on the concat) and Conv_1 + folded Conv_2
    # (64, 128 or — the cat(64, 128) block of the 128^2 up path — 192 raw channels through the centre tap) run on the
    # register-weight kernel: 16-row and 8-row tiles, one and
    # several tiles per block, FIR up / down in front (raw-input mode of Conv_0)
    cout = 64
    shp = dict(cin=(cin,), cout=(cout,), w0=(cout, cin, 3, 3), w1=(cout, cout, 3, 3), w2=(cout, cin, 1, 1), d=(cout, 64))
    tbl = [(n, shp[k]) for n, k in RB]
    sd = synth.synth_state_dict(tbl, 31)
    sd["Conv_2.weight"] = (sd["Conv_2.weight"] * 3.0).astype(np.float32)  # a wrong skip product cannot hide
    x, temb = rnd(f"rwrb.x{cin}{H}{W}{B}", (B, cin, H, W)), rnd("rwrb.t", (B, 64))
    ref = O._res_block(O.to_torch(sd), "", x, temb, up=up, down=down)
    yb = ops.resblock_forward([sd[n] for n, _ in tbl], ops.to_nhwc(x).to(DT).to(DEV), temb.to(DEV), cout, up=up, down=down)
    assert rel_rms(ops.to_nchw(yb).float(), ref) < 1.5e-2
