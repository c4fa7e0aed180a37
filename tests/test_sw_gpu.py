"""GPU parity of the streamed-weight 3x3 kernel (conv3x3_sw.hip: 64 .. 256 -> 128 / 256 couts, optional folded 1x1 skip on up
to 256 raw channels, 16-bit storage) through the C-ABI unit entry diffsep_conv3x3_streamed against torch fp32 on the CPU (same
16-bit-rounded operands), and of the engine's dispatch to it (whole residual blocks against the CPU oracle).
Tolerance: 4e-3 relative RMS per convolution (16-bit storage of the activated input and of the output; 6e-3 with bfloat16 weights
and a 256-channel skip), 1.5e-2 per residual block.  Reference layers: layers.py:141-156, layerspp.py:291-323, ncsnpp.py:409-417."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from diffsep_amd import _lib, ops, synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = "cuda"


def rel_rms(a, b):
    a = a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = b.detach().double().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / (np.sqrt(np.mean(b ** 2)) + 1e-30))


def rnd(tag, shape, scale=1.0):
    return torch.from_numpy(synth.synth_noise(tag, shape)) * scale


def test_frag_index_matches_the_library():
    """ops.pack_frag_weight restates include/diffsep_hip.h's formula: check it against diffsep_frag_index (host function)."""
    L = _lib.lib("f16")
    for (O, I, taps) in [(128, 64, 9), (128, 256, 9), (256, 192, 1)]:
        w = torch.arange(O * I * taps, dtype=torch.float32).reshape(O, I, 3 if taps == 9 else 1, 3 if taps == 9 else 1)
        p = ops.pack_frag_weight(w, torch.float32)
        for (o, t, i) in [(0, 0, 0), (5, taps - 1, 17), (O - 1, taps // 2, I - 1), (33, 0, 70 % I), (97, taps - 1, 63)]:
            k = L.diffsep_frag_index(o, t, i, taps, O)
            assert p[k] == w.reshape(O, I, taps)[o, i, t]
        # the split mode's copy: plane 0 = bf16(w), plane 1 = bf16(w - plane 0)
        w2 = (w + 0.37) / 1024.0
        ps = ops.pack_frag_weight_split(w2).float()
        for (o, t, i) in [(0, 0, 0), (5, taps - 1, 17), (O - 1, taps // 2, I - 1), (33, 0, 70 % I), (97, taps - 1, 63)]:
            k0 = L.diffsep_frag_index_split(o, t, i, taps, O, 0)
            k1 = L.diffsep_frag_index_split(o, t, i, taps, O, 1)
            v = w2.reshape(O, I, taps)[o, i, t]
            hi = v.to(torch.bfloat16).float()
            assert ps[k0] == hi and ps[k1] == (v - hi).to(torch.bfloat16).float()


# (B, H, W): one tile per block; several tiles per block and image borders inside a block's range; one row of tiles; wide
SHAPES = [(2, 8, 32), (3, 64, 96), (40, 64, 64), (1, 128, 256), (5, 24, 32), (3, 12, 64)]  # (H = 12: the 4 x 32 tile variant)
# (C1, C2, raw input?, skip channels (first, second), Cout)
CASES = [
    (64, 0, True, None, 128),        # 64 -> 128 behind a down-sampling (raw input)
    (128, 0, True, None, 128),
    (128, 0, False, None, 128),
    (128, 64, False, None, 128),     # cat(128, 64) -> 128
    (128, 128, False, None, 128),    # cat(128, 128) -> 128
    (128, 0, False, (64, 0), 128),
    (128, 0, False, (128, 0), 128),
    (128, 0, False, (128, 64), 128),   # Conv_1 + folded skip on cat(128, 64)
    (128, 0, False, (128, 128), 128),  # Conv_1 + folded skip on cat(128, 128)
    (128, 128, False, None, 256),    # two cout blocks
    (128, 0, False, "res", 128),     # Conv_1 + residual (against the identity copy)
    (128, 64, False, None, 64),      # cat(128, 64) -> 64: two cout groups x two pixel groups
    # nf = 128: 256-channel blocks (two cout blocks; 512 input channels: two table entries per thread, 16 channels per GroupNorm group)
    (256, 256, False, None, 256), (256, 128, False, None, 128), (256, 0, False, (256, 256), 256), (256, 0, False, (256, 128), 256),
    (256, 0, False, "res", 256), (128, 0, False, (256, 128), 128),
]


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,H,W", SHAPES)
@pytest.mark.parametrize("C1,C2,raw,skip,CO", CASES)
def test_sw_conv3x3_matches_torch(dt, B, H, W, C1, C2, raw, skip, CO):
    if dt == torch.bfloat16 and (B, H, W) not in ((2, 8, 32), (3, 64, 96)):
        pytest.skip("bfloat16 build: two shapes")
    if B * H * W > 200000 and (raw or CO == 256):
        pytest.skip("large case covered by the GroupNorm variants")
    if (C1 + C2 > 256 or C1 == 256) and (B, H, W) not in ((2, 8, 32), (3, 64, 96), (3, 12, 64)):
        pytest.skip("nf = 128 shapes: three image sizes")
    C = C1 + C2
    tag = f"{B}.{H}.{C1}.{C2}"
    a = (rnd("sw.a" + tag, (B, H, W, C1), 1.2) + 0.1).to(DEV, dt)
    bt = (rnd("sw.b" + tag, (B, H, W, C2), 0.9) - 0.2).to(DEV, dt) if C2 else None
    w = rnd(f"sw.w{C}.{CO}", (CO, C, 3, 3), 1.0 / math.sqrt(9 * C))
    bias, bb = rnd("sw.bias", (CO,), 0.1).to(DEV), rnd(f"sw.bb{B}", (B, CO), 0.1).to(DEV)
    sc = (1.0 + rnd(f"sw.sc{B}{C}", (B, C), 0.2)).to(DEV)
    sh = rnd(f"sw.sh{B}{C}", (B, C), 0.2).to(DEV)
    xf = torch.cat([a.float(), bt.float()], -1) if C2 else a.float()
    if not raw:
        xf = F.silu(xf * sc[:, None, None, :] + sh[:, None, None, :]).to(dt).float()
    ref = F.conv2d(xf.cpu().permute(0, 3, 1, 2), w.to(dt).float(), bias.cpu(), padding=1).permute(0, 2, 3, 1)
    ref = ref + bb.cpu()[:, None, None, :]
    sk, res, ident = None, None, None
    if CO == 64 and H % 8 != 0:
        pytest.skip("64 couts: 8-row tiles only")
    if skip == "res":
        res = rnd("sw.r" + tag, (B, H, W, CO), 1.3).to(DEV, dt)
        ident = ops.pack_frag_weight(torch.eye(CO).reshape(CO, CO, 1, 1), dt).to(DEV)
        ref = ref + res.float().cpu()
    elif skip is not None:
        s1, s2 = skip
        sa = rnd("sw.sa" + tag, (B, H, W, s1), 1.1).to(DEV, dt)
        sb = rnd("sw.sb" + tag, (B, H, W, s2), 0.8).to(DEV, dt) if s2 else None
        sw = rnd(f"sw.sw{s1 + s2}", (CO, s1 + s2, 1, 1), 1.0 / math.sqrt(s1 + s2))
        sxf = torch.cat([sa.float(), sb.float()], -1) if s2 else sa.float()
        ref = ref + F.conv2d(sxf.cpu().permute(0, 3, 1, 2), sw.to(dt).float(), None).permute(0, 2, 3, 1)
        sk = (sa, sb, ops.pack_frag_weight(sw, dt).to(DEV))
    ref = ref * 0.70710678
    y, st = ops.conv3x3_streamed(a, ops.pack_frag_weight(w, dt).to(DEV), CO, x2=bt, gn=None if raw else (sc, sh), bias=bias,
                                 bias_b=bb, skip=sk, out_scale=0.70710678, stats=True, res=res, ident_frag=ident)
    tol = 4e-3 if dt == torch.float16 else 8e-3
    assert rel_rms(y.float(), ref) < tol
    s = ops.stats_to_float(st)
    assert torch.allclose(s[..., 0].cpu(), ref.double().sum((1, 2)), rtol=3e-3, atol=3e-3 * H * W)
    assert torch.allclose(s[..., 1].cpu(), (ref.double() ** 2).sum((1, 2)), rtol=4e-3, atol=4e-3 * H * W)
    # plain launch: no bias / statistics
    y2 = ops.conv3x3_streamed(a, ops.pack_frag_weight(w, dt).to(DEV), CO, x2=bt, gn=None if raw else (sc, sh), skip=sk, res=res,
                              ident_frag=ident)
    ref2 = (ref / 0.70710678) - bias.cpu() - bb.cpu()[:, None, None, :]
    assert rel_rms(y2.float(), ref2) < tol
    # the same launch twice: bit-identical (fixed summation order, integer statistics)
    y3, st3 = ops.conv3x3_streamed(a, ops.pack_frag_weight(w, dt).to(DEV), CO, x2=bt, gn=None if raw else (sc, sh), bias=bias,
                                   bias_b=bb, skip=sk, out_scale=0.70710678, stats=True, res=res, ident_frag=ident)
    assert torch.equal(y, y3) and torch.equal(st, st3)


def test_sw_four_row_tiles_on_every_shape():
    """the 4 x 32 tile variant (the engine's choice below one 8-row tile per compute unit) forced on shapes the 8-row variant takes"""
    L = _lib.lib("f16")
    _lib.check(L.diffsep_set_option(b"sw_rows4", 1), L)
    try:
        for (B, H, W) in [(3, 64, 96), (2, 32, 32)]:
            for case in [(128, 128, False, None, 128), (128, 0, False, (128, 128), 128), (128, 0, False, "res", 128), (64, 0, True, None, 128)]:
                test_sw_conv3x3_matches_torch(torch.float16, B, H, W, *case)
    finally:
        _lib.check(L.diffsep_set_option(b"sw_rows4", 0), L)


def test_sw_rejects_what_it_does_not_instantiate():
    a = torch.zeros((1, 8, 32, 64), device=DEV, dtype=torch.float16)
    w = ops.pack_frag_weight(torch.zeros((64, 64, 3, 3)), torch.float16).to(DEV)
    with pytest.raises(RuntimeError):
        ops.conv3x3_streamed(a, w, 64)  # 64 -> 64: the register-weight kernel's layer
    a2 = torch.zeros((1, 10, 32, 128), device=DEV, dtype=torch.float16)
    w2 = ops.pack_frag_weight(torch.zeros((128, 128, 3, 3)), torch.float16).to(DEV)
    with pytest.raises(RuntimeError):
        ops.conv3x3_streamed(a2, w2, 128)  # H % 4 != 0


# ------------------------------------------------------------------------------------------------ split mode (conv3x3_sws.hip)
# (C1, C2, raw input?, skip channels (first, second) | "res", Cout)
SPLIT_CASES = [
    (64, 0, True, None, 64), (128, 0, True, None, 128),
    (64, 0, False, None, 64), (64, 64, False, None, 64), (128, 0, False, None, 64), (128, 64, False, None, 64),
    (128, 128, False, None, 128), (128, 64, False, None, 128), (128, 0, False, None, 128),
    (64, 0, False, "res", 64), (64, 0, False, (64, 0), 64), (64, 0, False, (64, 64), 64), (64, 0, False, (128, 64), 64),
    (128, 0, False, "res", 128), (128, 0, False, (64, 0), 128), (128, 0, False, (128, 0), 128), (128, 0, False, (128, 64), 128),
    (128, 0, False, (128, 128), 128), (128, 128, False, None, 256), (128, 128, True, None, 256),
]


@pytest.mark.parametrize("B,H,W", [(2, 8, 32), (3, 64, 96), (20, 64, 64), (5, 24, 32)])
@pytest.mark.parametrize("C1,C2,raw,skip,CO", SPLIT_CASES)
def test_sws_split_conv3x3_matches_torch_fp32(B, H, W, C1, C2, raw, skip, CO):
    """fp32 tensors, hi / lo bfloat16 products: 2^-17 relative per product -> 2e-5 relative RMS on the output (the generic tile in
    split mode has the same gate in tests/test_split_gpu.py)."""
    if B * H * W > 60000 and raw:
        pytest.skip("large case covered by the GroupNorm variants")
    C = C1 + C2
    tag = f"s{B}.{H}.{C1}.{C2}"
    a = (rnd("sws.a" + tag, (B, H, W, C1), 1.2) + 0.1).to(DEV)
    bt = (rnd("sws.b" + tag, (B, H, W, C2), 0.9) - 0.2).to(DEV) if C2 else None
    w = rnd(f"sws.w{C}.{CO}", (CO, C, 3, 3), 1.0 / math.sqrt(9 * C))
    bias, bb = rnd("sws.bias", (CO,), 0.1).to(DEV), rnd(f"sws.bb{B}", (B, CO), 0.1).to(DEV)
    sc = (1.0 + rnd(f"sws.sc{B}{C}", (B, C), 0.2)).to(DEV)
    sh = rnd(f"sws.sh{B}{C}", (B, C), 0.2).to(DEV)
    xf = torch.cat([a, bt], -1) if C2 else a
    if not raw:
        xf = F.silu(xf * sc[:, None, None, :] + sh[:, None, None, :])
    ref = F.conv2d(xf.double().cpu().permute(0, 3, 1, 2), w.double(), bias.double().cpu(), padding=1).permute(0, 2, 3, 1)
    ref = ref + bb.double().cpu()[:, None, None, :]
    sk, res, ident = None, None, None
    if skip == "res":
        res = rnd("sws.r" + tag, (B, H, W, CO), 1.3).to(DEV)
        ident = ops.pack_frag_weight_split(torch.eye(CO).reshape(CO, CO, 1, 1)).to(DEV)
        ref = ref + res.double().cpu()
    elif skip is not None:
        s1, s2 = skip
        sa = rnd("sws.sa" + tag, (B, H, W, s1), 1.1).to(DEV)
        sb = rnd("sws.sb" + tag, (B, H, W, s2), 0.8).to(DEV) if s2 else None
        sw = rnd(f"sws.sw{s1 + s2}", (CO, s1 + s2, 1, 1), 1.0 / math.sqrt(s1 + s2))
        sxf = torch.cat([sa, sb], -1) if s2 else sa
        ref = ref + F.conv2d(sxf.double().cpu().permute(0, 3, 1, 2), sw.double(), None).permute(0, 2, 3, 1)
        sk = (sa, sb, ops.pack_frag_weight_split(sw).to(DEV))
    ref = ref * 0.70710678
    wf = ops.pack_frag_weight_split(w).to(DEV)
    y, st = ops.conv3x3_streamed(a, wf, CO, x2=bt, gn=None if raw else (sc, sh), bias=bias, bias_b=bb, skip=sk, out_scale=0.70710678,
                                 stats=True, res=res, ident_frag=ident)
    r = rel_rms(y, ref)
    assert r < 2e-5, r
    s = ops.stats_to_float(st)
    assert torch.allclose(s[..., 0].cpu(), ref.sum((1, 2)), rtol=1e-4, atol=1e-4 * H * W)
    assert torch.allclose(s[..., 1].cpu(), (ref ** 2).sum((1, 2)), rtol=1e-4, atol=1e-4 * H * W)
    y3, st3 = ops.conv3x3_streamed(a, wf, CO, x2=bt, gn=None if raw else (sc, sh), bias=bias, bias_b=bb, skip=sk, out_scale=0.70710678,
                                   stats=True, res=res, ident_frag=ident)
    assert torch.equal(y, y3) and torch.equal(st, st3)


# ------------------------------------------------------------------------------------------------ the engine's dispatch
def _score_with_kernels(eng, xt, t, mn):
    eng.set_graph(False)
    eng.profile_begin()
    y = eng.score(xt, t, mn)
    eng.profile_end()
    return y, {r["kernel"] for r in eng.profile_records()}


@pytest.mark.parametrize("mode", ["f16", "split"])
def test_engine_dispatches_the_streamed_kernels_and_matches_the_oracle(mode):
    """nf = 64 at the bench's batch (B = 16 x 4 s): the 64- / 32-row levels and cat(128, 64) -> 64 run on conv3x3_sw.hip (8-row,
    4-row and 64-cout variants) in the half-precision engine, every >= 32-row 3x3 layer with >= 64 couts on conv3x3_sws.hip in the
    split engine; the options no_sw / no_sws restore the dispatch of before; both against the CPU oracle (one evaluation)."""
    import diffsep_oracle as O
    from diffsep_amd.engine import Engine, pack_state_dict, param_table
    dt = _lib.F16 if mode == "f16" else _lib.F32_SPLIT
    cfg = _lib.model_config(nf=64, num_sources=2, dtype=dt)
    sd = synth.synth_state_dict([(n, s) for n, s, _ in param_table(cfg)], 7)
    eng = Engine(cfg, pack_state_dict(cfg, sd))
    B, T = 16, 32000
    mix = torch.from_numpy(synth.synth_batch(B, T=T)[0])
    mn, _, _ = O.normalize_batch(mix)
    ocfg = O.default_config(64, 2)
    xt = O.prior_sampling(ocfg, mn, rnd("swe.z", (B, 2, T)))
    t = torch.full((B,), 0.5)
    torch.set_num_threads(min(torch.get_num_threads(), 16))
    ref = O.score_forward(O.to_torch(sd), ocfg, xt[:2], t[:2], mn[:2])  # (two utterances of the batch on the CPU)
    y, ks = _score_with_kernels(eng, xt.to(DEV), t.to(DEV), mn.to(DEV))
    name, opt = ("conv3x3_sw_kernel", "no_sw") if mode == "f16" else ("conv3x3_sws_kernel", "no_sws")
    mine = sorted(k for k in ks if k.startswith(name + "<"))
    print(f"\n[{mode}] streamed instantiations in one evaluation: {mine}")
    if mode == "f16":
        assert "conv3x3_sw_kernel<4,0,2,8,4>" in mine and "conv3x3_sw_kernel<4,0,2,4,4>" in mine and "conv3x3_sw_kernel<3,0,2,4,2>" in mine
        assert "conv3x3_sw_kernel<2,2,2,8,4>" in mine  # (Conv_1 + residual of the 64-row level: the identity copy)
    else:
        assert "conv3x3_sws_kernel<4,0,2,2>" in mine and "conv3x3_sws_kernel<8,0,2,4>" in mine and "conv3x3_sws_kernel<2,2,2,2>" in mine
    eng.set_option(opt, 1)
    y0, ks0 = _score_with_kernels(eng, xt.to(DEV), t.to(DEV), mn.to(DEV))
    eng.set_option(opt, 0)
    assert not any(k.startswith(name + "<") for k in ks0)
    tol = 8e-3 if mode == "f16" else 1e-4  # (a whole evaluation: ~150 layers of 2^-17 products)
    r, r0 = rel_rms(y[:2], ref), rel_rms(y0[:2], ref)
    print(f"[{mode}] one evaluation vs the CPU oracle: streamed {r:.3e}, option {opt} {r0:.3e}")
    assert r < tol and r0 < tol and not torch.equal(y, y0)
