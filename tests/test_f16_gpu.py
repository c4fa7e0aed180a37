"""GPU parity of the half-precision build (libdiffsep_hip_f16.so: the same kernels with IEEE fp16 instead of bfloat16 as
the 16-bit storage format, 11 instead of 8 significand bits) through the C-ABI: unit convolutions against torch fp32 on
the CPU, one full-size score evaluation against the CPU oracle, and the 60-NFE sampler against the fp32 engine.
Tolerances: fp16 storage rounds to 2^-11: 1e-3 per convolution (bf16: 4e-3), 5e-3 per score evaluation (bf16: 2.5e-2);
the sampler gate is SI-SDR against the fp32 engine's output on the same noise (bf16: 32 dB mean / 25 dB min)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import diffsep_oracle as O
from diffsep_amd import _lib, ops, synth
from diffsep_amd.engine import Engine, pack_state_dict, param_table

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = "cuda"
H16 = torch.float16
SDE = dict(ndim=2, d_lambda=2.0, sigma_min=0.05, sigma_max=0.5)


def rel_rms(a, b):
    a = a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = b.detach().double().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / (np.sqrt(np.mean(b ** 2)) + 1e-30))


def rnd(tag, shape, scale=1.0):
    return torch.from_numpy(synth.synth_noise(tag, shape)) * scale


def si_sdr(est, ref):
    est, ref = est.double().cpu(), ref.double().cpu()
    a = (est * ref).sum(-1, keepdim=True) / (ref * ref).sum(-1, keepdim=True)
    return 10 * torch.log10(((a * ref) ** 2).sum(-1) / ((est - a * ref) ** 2).sum(-1))


_ENG = {}


def engine(nf, dtype, seed=7):
    if (nf, dtype) not in _ENG:
        cfg = _lib.model_config(nf=nf, num_sources=2, dtype=dtype, spec_factor=0.33)
        sd = synth.synth_state_dict([(n, s) for n, s, _ in param_table(cfg)], seed)
        _ENG[(nf, dtype)] = (Engine(cfg, pack_state_dict(cfg, sd)), sd)
    return _ENG[(nf, dtype)]


# one shape per kernel family: generic tile, small-image kernel, register-weight kernel (64 and cat(64, 64)), weight-
# stationary kernel (residual), pyramid head, first layer, 1x1
@pytest.mark.parametrize("B,C1,C2,Cout,H,W,k,res", [(2, 128, 0, 128, 32, 32, 3, True), (2, 128, 128, 128, 8, 8, 3, False),
                                                    (3, 64, 0, 64, 64, 64, 3, False), (2, 64, 64, 64, 32, 64, 3, False),
                                                    (2, 64, 0, 64, 32, 64, 3, True), (2, 64, 0, 6, 16, 64, 3, False),
                                                    (2, 8, 0, 64, 16, 32, 3, False), (2, 128, 0, 64, 16, 32, 1, True)])
def test_f16_convolutions_match_torch(B, C1, C2, Cout, H, W, k, res):
    C = C1 + C2
    a = (rnd(f"h.a{C1}{H}", (B, H, W, C1), 1.2) + 0.1).to(DEV, H16)
    bt = (rnd(f"h.b{C2}{H}", (B, H, W, C2), 0.9) - 0.2).to(DEV, H16) if C2 else None
    w = rnd(f"h.w{C}{Cout}{k}", (Cout, C, k, k), 1.0 / math.sqrt(k * k * C))
    bias = rnd(f"h.bias{Cout}", (Cout,), 0.1).to(DEV)
    cp = (Cout + 7) // 8 * 8
    r = rnd(f"h.r{Cout}{H}", (B, H, W, cp)).to(DEV, H16) if res else None
    gn = C >= 64
    sc = (1.0 + rnd(f"h.sc{C}", (B, C), 0.2)).to(DEV)
    sh = rnd(f"h.sh{C}", (B, C), 0.2).to(DEV)
    xf = torch.cat([a.float(), bt.float()], -1) if C2 else a.float()
    if gn:
        xf = F.silu(xf * sc[:, None, None, :] + sh[:, None, None, :]).to(H16).float()
    ref = F.conv2d(xf.cpu().permute(0, 3, 1, 2), w.to(H16).float(), bias.cpu(), padding=k // 2).permute(0, 2, 3, 1)
    if res:
        ref = (ref + r.float().cpu()[..., :Cout]) * 0.70710678
    y, st = ops.conv2d_fused(a, ops.pack_conv_weight(w, H16).to(DEV), bias, Cout, k, x2=bt, gn=(sc, sh) if gn else None,
                             gn_act=1 if gn else 0, res=r, out_scale=0.70710678 if res else 1.0, cout_pad=cp, stats=True)
    assert y.dtype == H16
    assert rel_rms(y.float()[..., :Cout], ref) < 1e-3
    s = ops.stats_to_float(st)
    assert torch.allclose(s[..., 0].cpu(), ref.double().sum((1, 2)), rtol=5e-4, atol=5e-4 * H * W)


def test_f16_elementwise_kernels():
    # GroupNorm + SiLU with FIR up / down, channel concat and softmax on half-precision tensors
    B, H, W, C = 2, 16, 32, 64
    x = (rnd("h.gn", (B, H, W, C), 1.5) + 0.3).to(DEV, H16)
    g, be = (1.0 + rnd("h.g", (C,), 0.2)).to(DEV), rnd("h.be", (C,), 0.1).to(DEV)
    for resample in (0, 1, 2):
        y, xr = ops.groupnorm_act(x, g, be, 16, 1e-6, act=1, resample=resample, want_xr=True) if resample else \
            (ops.groupnorm_act(x, g, be, 16, 1e-6, act=1), None)
        xx = x.float().cpu().permute(0, 3, 1, 2)  # torch fp32 on the CPU
        hn = F.silu(F.group_norm(xx, 16, g.cpu(), be.cpu(), eps=1e-6))
        if resample == 1:
            hn, xx = O.fir_up2(hn), O.fir_up2(xx)
        elif resample == 2:
            hn, xx = O.fir_down2(hn), O.fir_down2(xx)
        assert rel_rms(ops.to_nchw(y).float(), hn) < 1.5e-3
        if resample:
            assert rel_rms(ops.to_nchw(xr).float(), xx) < 1.5e-3


@pytest.mark.parametrize("C,H,W", [(64, 64, 256), (128, 40, 256), (64, 256, 256), (128, 6, 1376), (192, 32, 192)])
def test_f16_fir_down_row_tiles(C, H, W):
    # round 5: the LDS row-tile FIR-down kernel of the half-precision build (W % 32 == 0, C % 64 == 0, >= 2^21 elements): whole
    # and ragged strips of 16 / 8 / 4 output rows, one / two / three 64-channel blocks, against torch fp32 on the CPU.  The
    # activation runs in packed half precision: same 1.5e-3 bar as the other half-precision elementwise kernels.
    B = 2
    groups = min(C // 4, 32)
    x = (rnd(f"h.fd{C}{H}", (B, H, W, C), 1.5) + 0.3).to(DEV, H16)
    g, be = (1.0 + rnd(f"h.fdg{C}", (C,), 0.2)).to(DEV), rnd(f"h.fdb{C}", (C,), 0.1).to(DEV)
    y, xr = ops.groupnorm_act(x, g, be, groups, 1e-6, act=1, resample=2, want_xr=True)
    xx = x.float().cpu().permute(0, 3, 1, 2)
    hn = O.fir_down2(F.silu(F.group_norm(xx, groups, g.cpu(), be.cpu(), eps=1e-6)))
    assert y.shape == (B, H // 2, W // 2, C) and xr.shape == y.shape
    assert rel_rms(ops.to_nchw(y).float(), hn) < 1.5e-3
    assert rel_rms(ops.to_nchw(xr).float(), O.fir_down2(xx)) < 1e-3
    # image borders: the first / last output rows and columns see the zero padding of BOTH tensors
    yb, hb = ops.to_nchw(y).float(), hn
    for sl in ((..., 0, slice(None)), (..., -1, slice(None)), (..., slice(None), 0), (..., slice(None), -1)):
        assert rel_rms(yb[sl], hb[sl]) < 2e-3


def test_f16_full_size_score_and_sampler():
    # one score evaluation at BASELINE's size against the CPU oracle, then the 60-NFE sampler against the fp32 engine
    T, N = 32000, 30
    eng16, sd = engine(64, _lib.F16)
    eng32, _ = engine(64, _lib.F32)
    assert eng16.kind == "f16" and eng32.kind == "bf16"
    cfg = O.default_config(64, 2)
    p = O.to_torch(sd)
    mix = torch.from_numpy(synth.synth_batch(1, T=T)[0])
    mixn, _, _ = O.normalize_batch(mix)
    xt = O.prior_sampling(cfg, mixn, rnd("h.z", (1, 2, T)))
    t = torch.tensor([0.7])
    ref = O.score_forward(p, cfg, xt, t, mixn)
    out = eng16.score(xt.to(DEV), t.to(DEV), mixn.to(DEV))
    r = rel_rms(out, ref)
    print(f"\n[f16 score evaluation, nf 64, T {T}] rel rms vs oracle {r:.3e}")
    assert r < 5e-3
    B8 = 8
    mix8 = torch.from_numpy(synth.synth_batch(B8, T=T)[0]).to(DEV)
    mn8, _, _ = ops.normalize_batch(mix8)
    kw = dict(N=N, corrector_steps=1, snr=0.5, eps=0.03, denoise=True, seed=99)
    ref8, _ = eng32.pc_sample(mn8, SDE, **kw)
    sep16, nfe = eng16.pc_sample(mn8, SDE, **kw)
    s = si_sdr(sep16, ref8)
    print(f"[f16 vs fp32 engine, {B8} utterances, {nfe} NFE] rel rms {rel_rms(sep16, ref8):.3e}  SI-SDR mean {float(s.mean()):.2f} "
          f"min {float(s.min()):.2f} dB")
    assert torch.isfinite(sep16).all()
    assert float(s.mean()) > 44.0 and float(s.min()) > 38.0
    # graph replay == eager, and a second call with the same seed is bit-identical
    again, _ = eng16.pc_sample(mn8, SDE, **kw)
    assert torch.equal(again, sep16)
