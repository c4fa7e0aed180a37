"""GPU parity of every HIP kernel (through the C-ABI) against the CPU oracle / torch fp32 ops and
the committed reference golden vectors.  Tolerances: fp32 kernels ~1e-5 relative RMS (summation
order only); bf16 kernels ~1e-2 (storage rounding), stated per test."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import diffsep_oracle as O
from diffsep_amd import ops, synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = "cuda"
SDE = dict(ndim=2, d_lambda=2.0, sigma_min=0.05, sigma_max=0.5)


def rel_rms(a, b):
    a = a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = b.detach().double().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / (np.sqrt(np.mean(b ** 2)) + 1e-30))


def rnd(tag, shape, scale=1.0):
    return torch.from_numpy(synth.synth_noise(tag, shape)) * scale


DT = [(torch.float32, 2e-5), (torch.bfloat16, 1.5e-2)]


@pytest.mark.parametrize("dtype,tol", DT)
@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 64, 64, 24, 64), (1, 128, 64, 16, 40), (2, 8, 64, 16, 32),
                                            (1, 64, 6, 16, 64), (2, 128, 128, 8, 8), (3, 128, 128, 4, 4),
                                            (1, 192, 128, 9, 33), (1, 256, 128, 16, 16)])
def test_conv3x3(dtype, tol, B, Cin, Cout, H, W):
    x = rnd(f"c3.x{Cin}{Cout}{H}", (B, Cin, H, W))
    w = rnd(f"c3.w{Cin}{Cout}", (Cout, Cin, 3, 3), 1.0 / math.sqrt(9 * Cin))
    b = rnd(f"c3.b{Cout}", (Cout,), 0.1)
    bb = rnd(f"c3.bb{Cout}", (B, Cout), 0.1)
    r = rnd(f"c3.r{Cout}{H}", (B, Cout, H, W))
    ref = (F.conv2d(x, w, b, padding=1) + bb[:, :, None, None] + r) / math.sqrt(2.0)
    cp = (Cout + 7) // 8 * 8
    y = ops.conv2d(ops.to_nhwc(x).to(DEV, dtype), ops.pack_conv_weight(w, dtype).to(DEV), b.to(DEV), Cout, 3,
                   bias_b=bb.to(DEV), res=ops.to_nhwc(r, cp).to(DEV, dtype), out_scale=1 / math.sqrt(2.0),
                   cout_pad=cp)
    assert rel_rms(ops.to_nchw(y.float(), Cout), ref) < tol
    if cp != Cout:
        assert float(y[..., Cout:].abs().max()) == 0.0  # channel padding is never written


@pytest.mark.parametrize("dtype,tol", DT)
@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 64, 128, 32, 64), (1, 8, 128, 16, 16), (2, 128, 128, 4, 4),
                                            (1, 8, 4, 32, 64), (1, 192, 64, 7, 9)])
def test_conv1x1(dtype, tol, B, Cin, Cout, H, W):
    x = rnd(f"c1.x{Cin}{Cout}{H}", (B, Cin, H, W))
    w = rnd(f"c1.w{Cin}{Cout}", (Cout, Cin, 1, 1), 1.0 / math.sqrt(Cin))
    b = rnd(f"c1.b{Cout}", (Cout,), 0.1)
    ref = F.conv2d(x, w, b)
    cp = (Cout + 7) // 8 * 8
    y = ops.conv2d(ops.to_nhwc(x).to(DEV, dtype), ops.pack_conv_weight(w, dtype).to(DEV), b.to(DEV), Cout, 1,
                   cout_pad=cp)
    assert rel_rms(ops.to_nchw(y.float(), Cout), ref) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("C1,C2,Cout,H,W,k", [(64, 64, 64, 16, 64, 3), (128, 64, 64, 16, 32, 3), (16, 16, 24, 8, 8, 3),
                                             (128, 128, 128, 4, 4, 3), (64, 64, 64, 16, 64, 1), (64, 0, 64, 24, 64, 3)])
def test_conv_with_fused_groupnorm_silu_and_concat(dtype, tol, C1, C2, Cout, H, W, k):
    # conv(silu(GN(cat([a, b])))) with the concat read in place and act(GN(.)) applied while staging:
    # the reference computes it as three separate ops (ncsnpp.py:411, layerspp.py:292-306)
    B = 2
    a = rnd(f"cf.a{C1}{H}", (B, C1, H, W), 1.3) + 0.2
    bt = rnd(f"cf.b{C2}{H}", (B, C2, H, W), 0.7) - 0.1 if C2 else None
    C = C1 + C2
    w = rnd(f"cf.w{C}{Cout}{k}", (Cout, C, k, k), 1.0 / math.sqrt(k * k * C))
    bias = rnd(f"cf.bias{Cout}", (Cout,), 0.1)
    g, be = 1.0 + rnd(f"cf.g{C}", (C,), 0.2), rnd(f"cf.be{C}", (C,), 0.1)
    groups = min(C // 4, 32)
    ar, br = a.to(dtype).float(), (bt.to(dtype).float() if C2 else None)
    xcat = torch.cat([ar, br], 1) if C2 else ar
    hn = F.silu(F.group_norm(xcat, groups, g, be, eps=1e-6))
    if dtype == torch.bfloat16:
        hn = hn.to(dtype).float()  # the kernel rounds act(GN(x)) to bf16 before the MFMA
    ref = F.conv2d(hn, w, bias, padding=k // 2)
    xa = ops.to_nhwc(a).to(DEV, dtype)
    xb = ops.to_nhwc(bt).to(DEV, dtype) if C2 else None
    sc, sh = ops.groupnorm_stats(xa, g.to(DEV), be.to(DEV), groups, 1e-6, x2=xb)
    y = ops.conv2d_fused(xa, ops.pack_conv_weight(w, dtype).to(DEV), bias.to(DEV), Cout, k, x2=xb, gn=(sc, sh),
                         gn_act=1)
    assert rel_rms(ops.to_nchw(y.float()), ref) < tol


def test_conv_f32_is_exact_fmaf_chain_on_integers():
    # small-integer operands: every product and partial sum is exactly representable, so the
    # MFMA result must be bit-identical to the CPU result regardless of summation order.
    x = torch.from_numpy(np.floor(synth.uniform01("ci.x", 2 * 64 * 16 * 32) * 7 - 3).astype(np.float32)).reshape(2, 64, 16, 32)
    w = torch.from_numpy(np.floor(synth.uniform01("ci.w", 64 * 64 * 9) * 5 - 2).astype(np.float32)).reshape(64, 64, 3, 3)
    ref = F.conv2d(x, w, None, padding=1)
    y = ops.conv2d(ops.to_nhwc(x).to(DEV), ops.pack_conv_weight(w, torch.float32).to(DEV), None, 64, 3)
    assert torch.equal(ops.to_nchw(y).cpu(), ref)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("C,H,W,resample", [(64, 32, 64, 0), (128, 16, 16, 0), (192, 8, 24, 0), (256, 4, 4, 0),
                                            (64, 16, 32, 1), (128, 16, 32, 2), (16, 8, 12, 1), (16, 8, 12, 2),
                                            # large levels: the LDS-tiled FIR-up kernel (whole and ragged tiles) and the
                                            # 2 x 2-block FIR-down kernel (fp32) / the strip FIR-down kernel (16-bit: whole and
                                            # ragged strips of 8 output rows, odd numbers of column pairs)
                                            (64, 100, 172, 1), (128, 256, 520, 2), (64, 128, 128, 1), (64, 256, 256, 2),
                                            (256, 44, 1032, 2)])
def test_groupnorm_silu_resample(dtype, tol, C, H, W, resample):
    B = 2
    x = rnd(f"gn.x{C}{H}{resample}", (B, C, H, W), 1.5) + 0.3
    g = 1.0 + rnd(f"gn.g{C}", (C,), 0.2)
    b = rnd(f"gn.b{C}", (C,), 0.1)
    groups = min(C // 4, 32)
    xin = x.to(dtype).float()  # the kernel sees the rounded input
    h = F.silu(F.group_norm(xin, groups, g, b, eps=1e-6))
    xr_ref = xin
    if resample == 1:
        h, xr_ref = O.fir_up2(h), O.fir_up2(xin)
    elif resample == 2:
        h, xr_ref = O.fir_down2(h), O.fir_down2(xin)
    y, xr = ops.groupnorm_act(ops.to_nhwc(x).to(DEV, dtype), g.to(DEV), b.to(DEV), groups, 1e-6, 1, resample,
                              want_xr=True)
    assert rel_rms(ops.to_nchw(y.float()), h) < tol
    if resample:
        assert rel_rms(ops.to_nchw(xr.float()), xr_ref) < tol


def test_groupnorm_without_activation_and_large_mean():
    # attention GroupNorm has no SiLU (layerspp.py:78); a large common offset stresses the variance
    x = rnd("gn2.x", (1, 128, 16, 16)) + 50.0
    g, b = torch.ones(128), torch.zeros(128)
    ref = F.group_norm(x, 32, g, b, eps=1e-6)
    y = ops.groupnorm_act(ops.to_nhwc(x).to(DEV), g.to(DEV), b.to(DEV), 32, 1e-6, 0, 0)
    assert rel_rms(ops.to_nchw(y), ref) < 2e-4


def test_upfirdn2d_matches_reference_golden(golden):
    g, _ = golden
    x = rnd("g1.x", (2, 8, 6, 10))
    up = ops.upfirdn2d(ops.to_nhwc(x).to(DEV), True)
    down = ops.upfirdn2d(ops.to_nhwc(x).to(DEV), False)
    assert np.abs(ops.to_nchw(up).cpu().numpy() - g["g1_up"]).max() < 1e-6
    assert np.abs(ops.to_nchw(down).cpu().numpy() - g["g1_down"]).max() < 1e-6


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("L,C", [(256, 128), (16, 128), (4, 16), (384, 256)])
def test_attention(dtype, tol, L, C):
    B = 2
    q, k, v = (rnd(f"at.{n}{L}{C}", (B, L, C)) for n in "qkv")
    qd, kd, vd = q.to(dtype).float(), k.to(dtype).float(), v.to(dtype).float()
    w = torch.softmax(torch.einsum("bic,bjc->bij", qd, kd) * C ** -0.5, dim=-1)
    ref = torch.einsum("bij,bjc->bic", w, vd)
    Lp = (L + 7) // 8 * 8
    vt = torch.zeros((B, C, Lp))
    vt[:, :, :L] = v.transpose(1, 2)
    o = ops.attention(q.to(DEV, dtype), k.to(DEV, dtype), vt.to(DEV, dtype))
    assert rel_rms(o.float(), ref) < tol


@pytest.mark.parametrize("T", [4000, 31999, 32000, 32001, 130])
def test_stft_pack_matches_oracle(T):
    cfg = O.default_config(16, 2)
    x = rnd(f"g6.x.{T}", (1, 3, T), 0.3)
    spec, _, n_pad = O.pre_process(cfg, x)
    W = spec.shape[-1]
    y = ops.stft_pack(x[:, :2].contiguous().to(DEV), x[:, 2:].contiguous().to(DEV), W, 8)
    got = ops.to_nchw(y, 6).cpu()
    assert rel_rms(got, spec) < 2e-5
    F_ = W - n_pad
    assert float(got[..., F_:].abs().max()) == 0.0 if n_pad else True
    # 2x - 1 variant (what the engine feeds the network), padded frames become exactly -1
    y2 = ops.to_nchw(ops.stft_pack(x[:, :2].contiguous().to(DEV), x[:, 2:].contiguous().to(DEV), W, 8, shift=True), 6).cpu()
    assert rel_rms(y2, 2 * spec - 1) < 2e-5
    if n_pad:
        assert torch.all(y2[..., F_:] == -1.0)


def test_stft_frame_indexing_bit_exact():
    # an impulse at sample n0 must appear in exactly the frames whose support [128 f - 255, 128 f + 255)
    # contains it — integer frame arithmetic identical to torch.stft(center=True) on the right-padded signal.
    T = 2000
    for n0 in (0, 1, 254, 255, 256, 1000, 1999):
        x = torch.zeros(1, 3, T)
        x[0, 0, n0] = 1.0
        W = 64
        y = ops.to_nchw(ops.stft_pack(x[:, :2].contiguous().to(DEV), x[:, 2:].contiguous().to(DEV), W, 8), 6).cpu()
        energy = (y[0, 0] ** 2 + y[0, 3] ** 2).sum(0)  # per frame
        F_ = 1 + (T + 382) // 128
        hit = [f for f in range(W) if energy[f] > 0]
        n_in = lambda f: n0 - (128 * f - 255)
        # the periodic Hann window is exactly 0 at tap 0, so tap index must be in [1, 509]
        want = [f for f in range(F_) if 1 <= n_in(f) <= 509]
        assert hit == want, (n0, hit, want)


def test_pre_process_matches_reference_golden(golden):
    g, _ = golden
    x = rnd("g6.x.4000", (1, 3, 4000), 0.3)
    y = ops.stft_pack(x[:, :2].contiguous().to(DEV), x[:, 2:].contiguous().to(DEV), 64, 8)
    assert rel_rms(ops.to_nchw(y, 6).cpu(), g["g6_pre_4000"]) < 2e-5


def test_istft_matches_reference_golden_and_roundtrip(golden):
    g, _ = golden
    yy = rnd("g6.y.4000", (1, 4, 256, 64), 0.2)
    out = ops.istft_unpack(ops.to_nhwc(yy, 8).to(DEV), 2, 4000)
    assert rel_rms(out.cpu(), g["g6_post_4000"]) < 2e-5
    # STFT -> iSTFT round trip through both kernels (compress o decompress = id): <= 2e-5
    T = 32000
    x = rnd("rt.x", (2, 3, T), 0.3)
    W = 256
    spec = ops.stft_pack(x[:, :2].contiguous().to(DEV), x[:, 2:].contiguous().to(DEV), W, 8)
    # take channels (re0, re1, im0, im1) -> [B,H,W,8] layout expected by the unpack (S=2 uses ch 0,1 | 2,3)
    sel = torch.zeros_like(spec)
    sel[..., 0], sel[..., 1], sel[..., 2], sel[..., 3] = spec[..., 0], spec[..., 1], spec[..., 3], spec[..., 4]
    back = ops.istft_unpack(sel, 2, T)
    assert rel_rms(back.cpu(), x[:, :2]) < 5e-5


def test_sde_updates_match_reference_golden(golden):
    g, _ = golden
    B, S, T, N = 2, 2, 4000, 3
    mix = torch.from_numpy(synth.synth_batch(B, T=T)[0])
    mix_norm, mean, std = ops.normalize_batch(mix.to(DEV))
    assert rel_rms(mix_norm.cpu(), g["g10_mix_norm"]) < 1e-6
    rm, rmean, rstd = O.normalize_batch(mix)
    assert rel_rms(mean.cpu(), rmean) < 1e-5 and rel_rms(std.cpu(), rstd) < 1e-6
    draws = [rnd(f"g9.z{i}", (B, S, T)) for i in range(3)]
    prior = ops.sde_prior(SDE, mix_norm, draws[0].to(DEV))
    assert rel_rms(prior.cpu(), g["g9_prior"]) < 1e-6
    # the score used by the golden updates comes from the reference network: recompute it with the oracle
    cfg = O.default_config(16, 2)
    p = O.to_torch(synth.synth_state_dict(O.param_table(cfg), 7))
    x0 = rnd("g9.x0", (B, S, T), 0.5)
    tv = torch.tensor([0.8, 0.2])
    sc = O.score_forward(p, cfg, x0, tv, torch.from_numpy(g["g10_mix_norm"]))
    xc, xcm = ops.sde_corrector_update(SDE, 0.5, x0.to(DEV), tv.to(DEV), sc.to(DEV), draws[1].to(DEV))
    xp, xpm = ops.sde_predictor_update(SDE, N, x0.to(DEV), tv.to(DEV), sc.to(DEV), draws[2].to(DEV))
    for a, k in ((xc, "g9_corr_x"), (xcm, "g9_corr_mean"), (xp, "g9_pred_x"), (xpm, "g9_pred_mean")):
        assert rel_rms(a.cpu(), g[k]) < 2e-5, k


def test_scale_output_matches_reference_golden(golden):
    g, _ = golden
    mix = torch.from_numpy(synth.synth_batch(2, T=4000)[0])
    out = ops.scale_output(mix.to(DEV), torch.from_numpy(g["g9_sep"]).to(DEV))
    assert rel_rms(out.cpu(), g["g10_scale_output"]) < 1e-6


def test_randn_is_standard_normal_and_reproducible():
    a = ops.randn(1 << 20, 123, 0)
    b = ops.randn(1 << 20, 123, 0)
    c = ops.randn(1 << 20, 123, 1)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert abs(float(a.mean())) < 5e-3 and abs(float(a.std()) - 1.0) < 5e-3
    assert abs(float((a ** 4).mean()) - 3.0) < 0.05
    assert abs(float((a * c).mean())) < 5e-3


def test_priormix_sde_kernels_match_reference_golden(golden):
    # PriorMixSDE (speech enhancement): time-varying std from the mixture envelope (sdes/sdes.py:352-590)
    g, _ = golden
    B, S, T, N = 2, 2, 4000, 3
    PSDE = dict(kind=1, ndim=2, d_lambda=2.0, sigma_min=0.05, sigma_max=0.5, avg_len=510)
    mix_norm = torch.from_numpy(g["g10_mix_norm"]).to(DEV)
    smix = ops.sde_sigma_mix(mix_norm, 510)
    assert rel_rms(smix.cpu(), g["g11_sigma_mix"][:, 0]) < 1e-6
    draws = [rnd(f"g9.z{i}", (B, S, T)) for i in range(3)]
    assert rel_rms(ops.sde_prior(PSDE, mix_norm, draws[0].to(DEV), smix).cpu(), g["g11_prior"]) < 1e-6
    cfg = O.default_config(16, 2)
    p = O.to_torch(synth.synth_state_dict(O.param_table(cfg), 7))
    x0 = rnd("g9.x0", (B, S, T), 0.5)
    tv = torch.tensor([0.8, 0.2])
    sc = O.score_forward(p, cfg, x0, tv, torch.from_numpy(g["g10_mix_norm"]))
    xc, xcm = ops.sde_corrector_update(PSDE, 0.5, x0.to(DEV), tv.to(DEV), sc.to(DEV), draws[1].to(DEV), smix)
    xp, xpm = ops.sde_predictor_update(PSDE, N, x0.to(DEV), tv.to(DEV), sc.to(DEV), draws[2].to(DEV), smix)
    for a, k in ((xc, "g11_corr_x"), (xcm, "g11_corr_mean"), (xp, "g11_pred_x"), (xpm, "g11_pred_mean")):
        assert rel_rms(a.cpu(), g[k]) < 2e-5, k
    # even window lengths drop the extra trailing sample; a window longer than the signal still works
    for k_len in (8, 509, 9000):
        ref = O.sigma_mix(torch.from_numpy(g["g10_mix_norm"]), k_len)[:, 0]
        assert rel_rms(ops.sde_sigma_mix(mix_norm, k_len).cpu(), ref) < 1e-5, k_len
    with pytest.raises(Exception):
        ops.sde_prior(PSDE, mix_norm, draws[0].to(DEV), None)  # PriorMixSDE without sigma_mix is an error


def test_remaining_correctors_and_predictors_match_reference_golden(golden):
    g, _ = golden
    B, S, T, N = 2, 2, 4000, 3
    mix_norm = torch.from_numpy(g["g10_mix_norm"])
    cfg = O.default_config(16, 2)
    p = O.to_torch(synth.synth_state_dict(O.param_table(cfg), 7))
    x0 = rnd("g9.x0", (B, S, T), 0.5)
    tv = torch.tensor([0.8, 0.2])
    sc = O.score_forward(p, cfg, x0, tv, mix_norm)
    draws = [rnd(f"g9.z{i}", (B, S, T)) for i in range(3)]
    xa, xam = ops.sde_corrector_update(SDE, 0.5, x0.to(DEV), tv.to(DEV), sc.to(DEV), draws[1].to(DEV), variant=1)
    assert rel_rms(xa.cpu(), g["g12_ald_x"]) < 2e-5 and rel_rms(xam.cpu(), g["g12_ald_mean"]) < 2e-5
    xl, xlm = ops.sde_langevin_update(0.5, x0.to(DEV), sc.to(DEV), draws[1].to(DEV))
    assert rel_rms(xl.cpu(), g["g12_langevin_x"]) < 2e-5 and rel_rms(xlm.cpu(), g["g12_langevin_mean"]) < 2e-5
    xe, xem = ops.sde_predictor_update(SDE, N, x0.to(DEV), tv.to(DEV), sc.to(DEV), draws[2].to(DEV))
    assert rel_rms(xe.cpu(), g["g12_em_x"]) < 2e-5 and rel_rms(xem.cpu(), g["g12_em_mean"]) < 2e-5
    xp, xpm = ops.sde_predictor_update(SDE, N, x0.to(DEV), tv.to(DEV), sc.to(DEV), draws[2].to(DEV),
                                       probability_flow=True)
    assert rel_rms(xpm.cpu(), g["g12_pflow_mean"]) < 2e-5 and torch.equal(xp, xpm)  # no noise on the ODE


def test_gram_and_separation_metrics():
    # SI-SDR / SI-SIR / SI-SAR with the best permutation from the HIP Gram kernel vs a float64 time-domain restatement
    from diffsep_amd import metrics
    B, S, T = 3, 2, 16000
    ref = rnd("met.ref", (B, S, T))
    mixm = torch.tensor([[0.9, 0.2], [0.1, 1.1]])
    est = torch.einsum("ij,bjt->bit", mixm, ref) + 0.05 * rnd("met.n", (B, S, T))
    est = est[:, [1, 0]]  # swapped: the permutation must be recovered
    G = ops.gram(ref.to(DEV), est.to(DEV)).cpu()
    assert torch.allclose(G[:, 0], torch.einsum("bit,bjt->bij", ref.double(), ref.double()), rtol=1e-9)
    assert torch.allclose(G[:, 1], torch.einsum("bit,bjt->bij", ref.double(), est.double()), rtol=1e-9, atol=1e-6)
    sdr, sir, sar, perm = metrics.si_bss_eval_sources(ref.to(DEV), est.to(DEV))
    assert (perm == np.array([1, 0])).all()
    r, e = ref.double().numpy(), est.double().numpy()[:, [1, 0]]
    for b in range(B):
        for i in range(S):
            tgt = (e[b, i] @ r[b, i]) / (r[b, i] @ r[b, i]) * r[b, i]
            coef = np.linalg.lstsq(r[b].T, e[b, i], rcond=None)[0]
            proj = coef @ r[b]
            assert abs(sdr[b, i] - 10 * np.log10((tgt @ tgt) / ((e[b, i] - tgt) @ (e[b, i] - tgt)))) < 1e-6
            assert abs(sir[b, i] - 10 * np.log10((tgt @ tgt) / ((proj - tgt) @ (proj - tgt)))) < 1e-5
            assert abs(sar[b, i] - 10 * np.log10((proj @ proj) / ((e[b, i] - proj) @ (e[b, i] - proj)))) < 1e-5
    # closed form: est = ref + orthogonal noise of relative power 1e-2 -> SI-SDR = 20 dB
    x = rnd("met.x", (1, 1, 8000)); n = rnd("met.o", (1, 1, 8000))
    n = n - (n * x).sum() / (x * x).sum() * x
    n = n * (0.1 * x.norm() / n.norm())
    sdr1, _, _, _ = metrics.si_bss_eval_sources(x.to(DEV), (x + n).to(DEV))
    assert abs(sdr1[0, 0] - 20.0) < 1e-3


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ksize,H,W", [(3, 16, 64), (3, 12, 40), (1, 16, 32), (3, 8, 8)])
def test_conv_epilogue_statistics(dt, ksize, H, W):
    # the channel-sum accumulators the conv epilogue fills for the next GroupNorm = statistics of its own output
    B, ci, co = 2, 16, 64
    x = rnd("cs.x", (B, H, W, ci)).to(DEV).to(dt)
    w = (rnd("cs.w", (co, ksize * ksize, ci)) / (ksize * ksize * ci) ** 0.5).to(DEV).to(dt)
    b = rnd("cs.b", (co,)).to(DEV)
    res = rnd("cs.r", (B, H, W, co)).to(DEV).to(dt)
    y, st = ops.conv2d_fused(x, w, b, co, ksize, res=res, out_scale=0.7071, stats=True)
    s = ops.stats_to_float(st)  # [B, co, 2]
    yd = y.double()
    tol = 2e-6 if dt == torch.float32 else 2e-3  # the sums are taken before the bf16 rounding of the output
    assert torch.allclose(s[..., 0], yd.sum((1, 2)), rtol=tol, atol=tol * H * W)
    assert torch.allclose(s[..., 1], (yd * yd).sum((1, 2)), rtol=tol, atol=tol * H * W)


@pytest.mark.parametrize("B,H,W", [(2, 16, 64), (5, 8, 32), (128, 16, 64), (256, 24, 32), (64, 48, 64)])
@pytest.mark.parametrize("act", [1, 0, None])
def test_weight_stationary_conv3x3_64_to_64(B, H, W, act):
    # the persistent 64 -> 64 bf16 kernel (conv3x3_ws.hip): one or several tiles per block, image borders,
    # GN affine (+SiLU) on the input, conv bias + per-batch temb bias, residual, 1/sqrt(2), statistics partials.
    # Reference: torch fp32 on the CPU (the same bf16-rounded operands)
    dt = torch.bfloat16
    x = (rnd(f"ws.x{B}{H}", (B, H, W, 64), 1.2) + 0.1).to(DEV).to(dt)
    w = rnd("ws.w", (64, 64, 3, 3), 1.0 / 24.0)
    bias, bb = rnd("ws.b", (64,), 0.1).to(DEV), rnd(f"ws.bb{B}", (B, 64), 0.1).to(DEV)
    res = rnd(f"ws.r{B}{H}", (B, H, W, 64)).to(DEV).to(dt)
    sc = (1.0 + rnd(f"ws.sc{B}", (B, 64), 0.2)).to(DEV)
    sh = rnd(f"ws.sh{B}", (B, 64), 0.2).to(DEV)
    xf = x.float()
    if act is not None:
        xf = xf * sc[:, None, None, :] + sh[:, None, None, :]
        if act:
            xf = F.silu(xf)
        xf = xf.to(dt).float()  # the kernel rounds the activated input to bf16 before the MFMA
    wq = w.to(dt).float()
    ref = F.conv2d(xf.cpu().permute(0, 3, 1, 2), wq, bias.cpu(), padding=1).permute(0, 2, 3, 1)
    ref = (ref + bb.cpu()[:, None, None, :] + res.float().cpu()) * 0.70710678
    y, st = ops.conv2d_fused(x, ops.pack_conv_weight(w, dt).to(DEV), bias, 64, 3, gn=None if act is None else (sc, sh),
                             gn_act=act or 0, bias_b=bb, res=res, out_scale=0.70710678, stats=True)
    assert rel_rms(y.float(), ref) < 4e-3
    s = ops.stats_to_float(st)
    assert torch.allclose(s[..., 0].cpu(), ref.double().sum((1, 2)), rtol=2e-3, atol=2e-3 * H * W)
    assert torch.allclose(s[..., 1].cpu(), (ref.double() ** 2).sum((1, 2)), rtol=2e-3, atol=2e-3 * H * W)
    y2 = ops.conv2d_fused(x, ops.pack_conv_weight(w, dt).to(DEV), None, 64, 3)  # no bias / residual / statistics
    ref2 = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), wq, None, padding=1).permute(0, 2, 3, 1)
    assert rel_rms(y2.float(), ref2) < 4e-3


@pytest.mark.parametrize("dtype,tol", DT)
@pytest.mark.parametrize("C1,C2,Cout,H,W,k", [(64, 0, 64, 16, 64, 3), (64, 64, 64, 16, 32, 3), (128, 64, 32, 8, 32, 3),
                                             (128, 0, 64, 16, 32, 1), (64, 64, 128, 4, 4, 3)])
def test_conv_chunk_major_weights_are_equivalent(dtype, tol, C1, C2, Cout, H, W, k):
    # chunk-major weights [Cin/kc][taps][Cout][kc] (the engine's layout) give bit-identical results to [Cout][taps][Cin]
    B = 2
    a = rnd(f"ck.a{C1}{H}", (B, H, W, C1)).to(DEV, dtype)
    bt = rnd(f"ck.b{C2}{H}", (B, H, W, C2)).to(DEV, dtype) if C2 else None
    w = rnd(f"ck.w{C1}{C2}{Cout}{k}", (Cout, C1 + C2, k, k), 1.0 / math.sqrt(k * k * (C1 + C2)))
    bias = rnd(f"ck.bias{Cout}", (Cout,), 0.1).to(DEV)
    kc = ops.conv2d_chunk(k, dtype)
    y0 = ops.conv2d_fused(a, ops.pack_conv_weight(w, dtype).to(DEV), bias, Cout, k, x2=bt)
    y1 = ops.conv2d_fused(a, ops.pack_conv_weight(w, dtype, chunk=kc).to(DEV), bias, Cout, k, x2=bt, w_chunk=kc)
    assert torch.equal(y0, y1)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("C1,C2,Cout,H,W,k", [(64, 0, 64, 16, 64, 3), (64, 64, 64, 16, 32, 3), (128, 64, 64, 8, 32, 3),
                                             (16, 16, 24, 8, 8, 3), (64, 64, 64, 16, 64, 1)])
def test_conv_groupnorm_from_producer_accumulators(dtype, tol, C1, C2, Cout, H, W, k, monkeypatch):
    # producer convs fill the channel-sum accumulators; the consumer derives GroupNorm scale / shift from them in its
    # own prologue (no finalize launch), also across the seam of an in-place concat (groups of 6 for C = 192)
    B = 2
    C = C1 + C2
    groups = min(C // 4, 32)
    g, be = (1.0 + rnd(f"ga.g{C}", (C,), 0.2)).to(DEV), rnd(f"ga.be{C}", (C,), 0.1).to(DEV)

    def produce(tag, Cp):
        xi = rnd(f"ga.x{tag}{Cp}{H}", (B, H, W, 16)).to(DEV, dtype)
        wi = ops.pack_conv_weight(rnd(f"ga.w{tag}{Cp}", (Cp, 16, 3, 3), 1.0 / 12.0), dtype).to(DEV)
        return ops.conv2d_fused(xi, wi, rnd(f"ga.b{tag}{Cp}", (Cp,), 0.3).to(DEV), Cp, 3, stats=True)

    a, sa = produce("a", C1)
    (bt, sb) = produce("b", C2) if C2 else (None, None)
    w = rnd(f"ga.w{C}{Cout}{k}", (Cout, C, k, k), 1.0 / math.sqrt(k * k * C))
    bias = rnd(f"ga.bias{Cout}", (Cout,), 0.1).to(DEV)
    wp = ops.pack_conv_weight(w, dtype).to(DEV)
    sc, sh = ops.groupnorm_stats(a, g, be, groups, 1e-6, x2=bt)  # two-pass statistics over the stored tensor
    y_ref = ops.conv2d_fused(a, wp, bias, Cout, k, x2=bt, gn=(sc, sh), gn_act=1)
    y = ops.conv2d_fused(a, wp, bias, Cout, k, x2=bt, gn_acc=(sa, sb, g, be, groups), gn_act=1)
    # (the accumulators hold the sums BEFORE the output was rounded to the storage dtype)
    assert rel_rms(y.float(), y_ref.float()) < (1e-5 if dtype == torch.float32 else 1e-2)
    # torch fp32 on the CPU (not MIOpen on the GPU)
    xcat = (torch.cat([a.float(), bt.float()], -1) if C2 else a.float()).cpu()
    hn = F.silu(F.group_norm(xcat.permute(0, 3, 1, 2), groups, g.cpu(), be.cpu(), eps=1e-6))
    if dtype == torch.bfloat16:
        hn = hn.to(dtype).float()
    ref = F.conv2d(hn, w, bias.cpu(), padding=k // 2).permute(0, 2, 3, 1)
    assert rel_rms(y.float(), ref) < tol


@pytest.mark.parametrize("C1,C2,Cout,H,W", [(128, 0, 128, 16, 16), (128, 0, 128, 8, 8), (128, 0, 128, 4, 4),
                                            (128, 128, 128, 8, 8), (64, 192, 48, 16, 12), (256, 256, 256, 4, 1),
                                            (64, 0, 16, 16, 4), (128, 64, 64, 8, 6), (128, 128, 128, 16, 24)])
@pytest.mark.parametrize("act,lazy", [(1, False), (1, True), (0, False), (None, False)])
def test_small_image_conv3x3(C1, C2, Cout, H, W, act, lazy):
    # the 16-cout-slab kernel of the <= 16-row levels (conv3x3_small.hip): whole and ragged tiles, concat views,
    # GroupNorm (+SiLU) from materialised tables or from the producers' accumulators, bias + per-sample bias, residual,
    # scale, statistics of the output; chunk-major and row-major weights.  Reference: torch fp32 on the CPU
    dt = torch.bfloat16
    B, C = 3, C1 + C2
    tag = f"{C1}.{C2}.{H}.{W}"
    xa = (rnd("sm.a" + tag, (B, H, W, C1), 1.2) + 0.1).to(DEV).to(dt)
    xb = (rnd("sm.b" + tag, (B, H, W, C2), 0.8) - 0.2).to(DEV).to(dt) if C2 else None
    w = rnd(f"sm.w{C}.{Cout}", (Cout, C, 3, 3), (9 * C) ** -0.5)
    bias, bb = rnd(f"sm.bias{Cout}", (Cout,), 0.1).to(DEV), rnd(f"sm.bb{Cout}", (B, Cout), 0.1).to(DEV)
    res = rnd(f"sm.r{Cout}" + tag, (B, H, W, Cout)).to(DEV).to(dt)
    groups = min(C // 4, 32)
    g, be = (1.0 + rnd(f"sm.g{C}", (C,), 0.2)).to(DEV), rnd(f"sm.be{C}", (C,), 0.1).to(DEV)
    xcat = torch.cat([xa, xb], -1).float().cpu() if C2 else xa.float().cpu()
    kw = {}
    if act is None:
        hn = xcat
    elif lazy:
        # accumulators as a producing conv would leave them: fixed-point sums of the stored tensors
        def acc(x):
            xd = x.double()
            return torch.stack([(xd.sum((1, 2)) * ops.STAT_SUM_SCALE).round(), ((xd * xd).sum((1, 2)) * ops.STAT_SQ_SCALE).round()],
                               -1).to(torch.int64).contiguous()
        kw = dict(gn_acc=(acc(xa), acc(xb) if C2 else None, g, be, groups), gn_act=act)
        hn = F.group_norm(xcat.permute(0, 3, 1, 2), groups, g.cpu(), be.cpu(), eps=1e-6).permute(0, 2, 3, 1)
    else:
        sc, sh = (1.0 + rnd(f"sm.sc{C}", (B, C), 0.2)).to(DEV), rnd(f"sm.sh{C}", (B, C), 0.2).to(DEV)
        kw = dict(gn=(sc, sh), gn_act=act)
        hn = xcat * sc.cpu()[:, None, None, :] + sh.cpu()[:, None, None, :]
    if act:
        hn = F.silu(hn)
    hn = hn.to(dt).float()  # the kernel rounds the activated input to bf16 before the MFMA
    ref = F.conv2d(hn.permute(0, 3, 1, 2), w.to(dt).float(), bias.cpu(), padding=1).permute(0, 2, 3, 1)
    ref = (ref + bb.cpu()[:, None, None, :] + res.float().cpu()) * 0.70710678
    for chunk in (0, 32):
        wp = ops.pack_conv_weight(w, dt, chunk=chunk).to(DEV)
        y, st = ops.conv2d_fused(xa, wp, bias, Cout, 3, x2=xb, bias_b=bb, res=res, out_scale=0.70710678, stats=True,
                                 w_chunk=chunk, **kw)
        assert rel_rms(y.float().cpu(), ref) < 4e-3
        s = ops.stats_to_float(st).cpu()
        assert torch.allclose(s[..., 0], ref.double().sum((1, 2)), rtol=2e-3, atol=2e-3 * H * W)
        assert torch.allclose(s[..., 1], (ref.double() ** 2).sum((1, 2)), rtol=2e-3, atol=2e-3 * H * W)
    y2 = ops.conv2d_fused(xa, ops.pack_conv_weight(w, dt).to(DEV), None, Cout, 3, x2=xb)  # bare convolution
    ref2 = F.conv2d(xcat.permute(0, 3, 1, 2), w.to(dt).float(), None, padding=1).permute(0, 2, 3, 1)
    assert rel_rms(y2.float().cpu(), ref2) < 4e-3


@pytest.mark.parametrize("C,Cout,H,W", [(128, 6, 4, 4), (128, 6, 8, 8), (128, 6, 16, 16), (128, 8, 16, 12), (256, 4, 8, 6)])
@pytest.mark.parametrize("chunk", [0, 32])
def test_small_image_pyramid_head_conv3x3(C, Cout, H, W, chunk):
    # the output-pyramid heads of the <= 16-row levels (<= 8 couts, GroupNorm + SiLU on the input, + the upsampled pyramid) on the
    # small-image kernel: one 16-cout slab whose couts past Cout multiply zeros and whose channel quads past the padded output
    # are neither read nor written; the padding channels of the output stay zero
    dt = torch.bfloat16
    B, cp = 3, 8
    tag = f"{C}.{Cout}.{H}.{W}"
    x = (rnd("hd.x" + tag, (B, H, W, C), 1.1) + 0.1).to(DEV).to(dt)
    w = rnd("hd.w" + tag, (Cout, C, 3, 3), (9 * C) ** -0.5)
    bias = rnd(f"hd.b{Cout}", (Cout,), 0.1).to(DEV)
    res = torch.zeros(B, H, W, cp)
    res[..., :Cout] = rnd("hd.r" + tag, (B, H, W, Cout))
    res = res.to(DEV).to(dt)
    sc, sh = (1.0 + rnd(f"hd.sc{C}", (B, C), 0.2)).to(DEV), rnd(f"hd.sh{C}", (B, C), 0.2).to(DEV)
    hn = F.silu(x.float().cpu() * sc.cpu()[:, None, None, :] + sh.cpu()[:, None, None, :]).to(dt).float()
    ref = F.conv2d(hn.permute(0, 3, 1, 2), w.to(dt).float(), bias.cpu(), padding=1).permute(0, 2, 3, 1) + res.float().cpu()[..., :Cout]
    wp = ops.pack_conv_weight(w, dt, chunk=chunk).to(DEV)
    y = torch.full((B, H, W, cp), 7.0, device=DEV, dtype=dt)
    y[..., Cout:] = 0  # (the engine's arena is zeroed: padding channels are never written)
    ops.conv2d_fused(x, wp, bias, Cout, 3, gn=(sc, sh), gn_act=1, res=res, cout_pad=cp, out=y, w_chunk=chunk)
    assert rel_rms(y[..., :Cout].float().cpu(), ref) < 4e-3
    assert Cout == cp or float(y[..., Cout:].abs().max()) == 0.0


@pytest.mark.parametrize("B,H,W,cin", [(2, 16, 32, 6), (16, 64, 64, 6), (3, 8, 96, 8), (64, 24, 32, 4)])
def test_first_layer_conv3x3_8_to_64(B, H, W, cin):
    # the network's first layer (<= 8 input channels, padded to 8) on its own persistent kernel in bf16
    # (conv3x3_thin_in_kernel): image borders, one and several tiles per block, bias, statistics of the output
    dt = torch.bfloat16
    x = torch.zeros(B, H, W, 8)
    x[..., :cin] = rnd(f"fl.x{B}{H}{W}", (B, H, W, cin), 1.3)
    x = x.to(DEV).to(dt)
    w = torch.zeros(64, 8, 3, 3)
    w[:, :cin] = rnd(f"fl.w{cin}", (64, cin, 3, 3), (9 * cin) ** -0.5)
    bias = rnd("fl.b", (64,), 0.2).to(DEV)
    y, st = ops.conv2d_fused(x, ops.pack_conv_weight(w, dt).to(DEV), bias, 64, 3, stats=True)
    ref = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.to(dt).float(), bias.cpu(), padding=1).permute(0, 2, 3, 1)
    assert rel_rms(y.float().cpu(), ref) < 4e-3
    s = ops.stats_to_float(st).cpu()
    assert torch.allclose(s[..., 0], ref.double().sum((1, 2)), rtol=2e-3, atol=2e-3 * H * W)
    assert torch.allclose(s[..., 1], (ref.double() ** 2).sum((1, 2)), rtol=2e-3, atol=2e-3 * H * W)


@pytest.mark.parametrize("Cin,Cout,B,H,W", [(64, 6, 2, 16, 32), (64, 6, 16, 64, 64), (128, 6, 3, 8, 64), (256, 4, 2, 8, 32),
                                           (64, 8, 2, 24, 96)])
@pytest.mark.parametrize("act,lazy,with_res", [(1, True, True), (1, False, False), (0, False, True)])
def test_pyramid_head_conv3x3_to_8_channels(Cin, Cout, B, H, W, act, lazy, with_res):
    # the output-pyramid heads (<= 8 couts) on their own kernel in bf16 (conv3x3_thin_out_kernel): GroupNorm (+ SiLU) from
    # tables or from the producer's accumulators, one or several 64-channel passes, residual, zero channel padding
    dt = torch.bfloat16
    tag = f"{Cin}.{Cout}.{H}.{W}"
    x = (rnd("ph.x" + tag, (B, H, W, Cin), 1.2) + 0.1).to(DEV).to(dt)
    w = rnd("ph.w" + tag, (Cout, Cin, 3, 3), (9 * Cin) ** -0.5)
    bias = rnd("ph.b" + tag, (Cout,), 0.1).to(DEV)
    res = torch.zeros(B, H, W, 8)
    res[..., :Cout] = rnd("ph.r" + tag, (B, H, W, Cout))
    res = res.to(DEV).to(dt)
    groups = min(Cin // 4, 32)
    g, be = (1.0 + rnd(f"ph.g{Cin}", (Cin,), 0.2)).to(DEV), rnd(f"ph.be{Cin}", (Cin,), 0.1).to(DEV)
    xf = x.float().cpu()
    if lazy:
        xd = x.double()
        acc = torch.stack([(xd.sum((1, 2)) * ops.STAT_SUM_SCALE).round(), ((xd * xd).sum((1, 2)) * ops.STAT_SQ_SCALE).round()],
                          -1).to(torch.int64).contiguous()
        kw = dict(gn_acc=(acc, None, g, be, groups), gn_act=act)
        hn = F.group_norm(xf.permute(0, 3, 1, 2), groups, g.cpu(), be.cpu(), eps=1e-6).permute(0, 2, 3, 1)
    else:
        sc, sh = (1.0 + rnd(f"ph.sc{Cin}", (B, Cin), 0.2)).to(DEV), rnd(f"ph.sh{Cin}", (B, Cin), 0.2).to(DEV)
        kw = dict(gn=(sc, sh), gn_act=act)
        hn = xf * sc.cpu()[:, None, None, :] + sh.cpu()[:, None, None, :]
    if act:
        hn = F.silu(hn)
    hn = hn.to(dt).float()
    ref = F.conv2d(hn.permute(0, 3, 1, 2), w.to(dt).float(), bias.cpu(), padding=1).permute(0, 2, 3, 1)
    if with_res:
        ref = ref + res.float().cpu()[..., :Cout]
    for chunk in (0, 32):
        y = ops.conv2d_fused(x, ops.pack_conv_weight(w, dt, chunk=chunk).to(DEV), bias, Cout, 3, res=res if with_res else None,
                             cout_pad=8, w_chunk=chunk, **kw)
        assert rel_rms(y.float().cpu()[..., :Cout], ref) < 4e-3
        if Cout < 8:
            assert not bool(y[..., Cout:].any())


@pytest.mark.parametrize("Cin,Cout,H,W", [(64, 128, 128, 64), (128, 256, 128, 64), (128, 64, 64, 64), (192, 64, 40, 72),
                                          (128, 64, 128, 128)])
def test_wide_tile_conv3x3_128_couts(Cin, Cout, H, W):
    # the tile choice of the generic bf16 kernel by launch size: >= 1024 blocks and 128 | Cout run on 128-cout tiles (8 waves,
    # two epilogue passes), 129 ... 1023 blocks on the standard 8 x 32 x 64-cout tile (the small-batch unit tests above all
    # land on the half-width tile), weight-heavy 64-cout launches of >= 1024 blocks on 16 x 32-pixel tiles.  GroupNorm + SiLU on the input, bias + per-sample bias, residual, scale, statistics;
    # reference = torch fp32 on the CPU
    dt = torch.bfloat16
    B = 16
    x = (rnd(f"wt.x{Cin}", (B, H, W, Cin), 1.2) + 0.1).to(DEV).to(dt)
    w = rnd(f"wt.w{Cin}{Cout}", (Cout, Cin, 3, 3), (9 * Cin) ** -0.5)
    bias, bb = rnd(f"wt.b{Cout}", (Cout,), 0.1).to(DEV), rnd(f"wt.bb{Cout}", (B, Cout), 0.1).to(DEV)
    res = rnd(f"wt.r{Cout}", (B, H, W, Cout)).to(DEV).to(dt)
    sc, sh = (1.0 + rnd(f"wt.sc{Cin}", (B, Cin), 0.2)).to(DEV), rnd(f"wt.sh{Cin}", (B, Cin), 0.2).to(DEV)
    hn = F.silu(x.float().cpu() * sc.cpu()[:, None, None, :] + sh.cpu()[:, None, None, :]).to(dt).float()
    ref = F.conv2d(hn.permute(0, 3, 1, 2), w.to(dt).float(), bias.cpu(), padding=1).permute(0, 2, 3, 1)
    ref = (ref + bb.cpu()[:, None, None, :] + res.float().cpu()) * 0.70710678
    y, st = ops.conv2d_fused(x, ops.pack_conv_weight(w, dt, chunk=32).to(DEV), bias, Cout, 3, gn=(sc, sh), gn_act=1,
                             bias_b=bb, res=res, out_scale=0.70710678, stats=True, w_chunk=32)
    assert rel_rms(y.float().cpu(), ref) < 4e-3
    s = ops.stats_to_float(st).cpu()
    assert torch.allclose(s[..., 0], ref.double().sum((1, 2)), rtol=2e-3, atol=2e-3 * H * W)
    assert torch.allclose(s[..., 1], (ref.double() ** 2).sum((1, 2)), rtol=2e-3, atol=2e-3 * H * W)
