"""DIFFSEP_F32_SPLIT — fp32 tensors, every MFMA product as three bf16 MFMAs on the hi / lo bf16 halves of both operands
(hi*hi + hi*lo + lo*hi, fp32 accumulation) — against the same references as the exact fp32 engine: the committed
reference golden vectors, torch fp32 on the CPU for single launches, and the CPU oracle at BASELINE.json's full size.
The bar is the north_star's: separated waveforms within 1e-3 RMS of the reference on identical inputs."""
import math

import pytest
import torch
import torch.nn.functional as F

import diffsep_oracle as O
from diffsep_amd import _lib, ops, synth
from test_engine_gpu import DEV, SDE, _g9_inputs, diff_rms, engine, rel_rms, rms, rnd, si_sdr

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.mark.parametrize("C1,C2,Cout,H,W,k", [(64, 0, 64, 16, 64, 3), (64, 64, 64, 16, 32, 3), (128, 64, 32, 8, 32, 3),
                                             (128, 0, 128, 8, 8, 3), (16, 16, 24, 12, 40, 3), (8, 0, 64, 16, 32, 3),
                                             (64, 64, 128, 16, 32, 1), (256, 0, 64, 4, 4, 1), (24, 0, 40, 8, 8, 1)])
def test_split_conv_launches_vs_torch_fp32(C1, C2, Cout, H, W, k):
    # every tile configuration of the generic kernel in split mode: GN + SiLU on the input, concat view, bias, per-sample
    # bias, residual, scale, statistics; chunk-major and row-major weights.  fp32 reference on the CPU; the exact fp32
    # kernel on the same inputs sets the scale of what "equal" means (summation order: ~1e-6)
    B, C = 2, C1 + C2
    tag = f"{C1}.{C2}.{Cout}.{H}.{W}.{k}"
    xa = (rnd("sp.a" + tag, (B, H, W, C1), 1.2) + 0.1).to(DEV)
    xb = (rnd("sp.b" + tag, (B, H, W, C2), 0.8) - 0.2).to(DEV) if C2 else None
    w = rnd("sp.w" + tag, (Cout, C, k, k), (k * k * C) ** -0.5)
    bias, bb = rnd("sp.bias" + tag, (Cout,), 0.1).to(DEV), rnd("sp.bb" + tag, (B, Cout), 0.1).to(DEV)
    res = rnd("sp.r" + tag, (B, H, W, Cout)).to(DEV)
    sc, sh = (1.0 + rnd("sp.sc" + tag, (B, C), 0.2)).to(DEV), rnd("sp.sh" + tag, (B, C), 0.2).to(DEV)
    xcat = torch.cat([xa, xb], -1).cpu() if C2 else xa.cpu()
    hn = F.silu(xcat * sc.cpu()[:, None, None, :] + sh.cpu()[:, None, None, :])
    ref = F.conv2d(hn.permute(0, 3, 1, 2).double(), w.double(), bias.cpu().double(), padding=k // 2).permute(0, 2, 3, 1)
    ref = ((ref + bb.cpu().double()[:, None, None, :] + res.cpu().double()) * 0.70710678).float()
    kc = ops.conv2d_chunk(k, torch.float32)
    for chunk in ((0, kc) if C % 64 == 0 and C1 % kc == 0 else (0,)):
        wp = ops.pack_conv_weight(w, torch.float32, chunk=chunk).to(DEV)
        kw = dict(x2=xb, gn=(sc, sh), gn_act=1, bias_b=bb, res=res, out_scale=0.70710678, stats=True, w_chunk=chunk)
        y, st = ops.conv2d_fused(xa, wp, bias, Cout, k, split=True, **kw)
        y_exact, _ = ops.conv2d_fused(xa, wp, bias, Cout, k, **kw)
        assert rel_rms(y_exact, ref) < 2e-6
        assert rel_rms(y, ref) < 2e-5, "bf16x3 products: 2^-17 per operand"
        s = ops.stats_to_float(st).cpu()
        assert torch.allclose(s[..., 0], ref.double().sum((1, 2)), rtol=1e-4, atol=1e-4 * H * W)


def test_split_attention_vs_torch():
    B, L, C = 2, 256, 128
    q, k, v = rnd("spa.q", (B, L, C)), rnd("spa.k", (B, L, C)), rnd("spa.v", (B, L, C))
    ref = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(C), -1) @ v
    vt = v.transpose(1, 2).contiguous()
    o = ops.attention(q.to(DEV), k.to(DEV), vt.to(DEV), split=True)
    assert rel_rms(o, ref) < 3e-5


def test_split_engine_matches_reference_golden(golden):
    g, meta = golden
    eng, _ = engine(16, 2, _lib.F32_SPLIT)
    T = 4000
    xt, mix = rnd("g7.xt", (2, 2, T), 0.5), rnd("g7.mix", (2, 1, T), 0.5)
    out = eng.score(xt.to(DEV), torch.tensor([0.7, 0.05], device=DEV), mix.to(DEV))
    assert rel_rms(out, g["g7_score"]) < 2e-4
    mixb, draws, N, cs = _g9_inputs()
    mix_norm, _, _ = ops.normalize_batch(mixb.to(DEV))
    sep, nfe = eng.pc_sample(mix_norm, SDE, N=N, corrector_steps=cs, snr=0.5, eps=0.03, denoise=True, noise=draws.to(DEV))
    assert nfe == meta["g9_nfe"]
    assert diff_rms(sep, g["g9_sep"]) < 1e-3 and rel_rms(sep, g["g9_sep"]) < 3e-4


def test_split_full_size_sampler_nf64_N30_parity_with_oracle(oracle_fullsize_nf64):
    # the parity gate of the fp32 engine (test_engine_gpu.py) for the split engine: 4 s / 8 kHz / 2 speakers / 60 NFE,
    # identical noise, against the CPU oracle (one oracle run per session: conftest.oracle_fullsize_nf64); and against the
    # exact fp32 engine (SI-SDR: what a listener could tell)
    fs = oracle_fullsize_nf64
    T, B, N, mix, draws, ref, nfe = fs["T"], fs["B"], fs["N"], fs["mix"], fs["draws"], fs["ref"], fs["nfe"]
    eng, sd = engine(64, 2, _lib.F32_SPLIT)
    mix_norm, _, _ = ops.normalize_batch(mix.to(DEV))
    kw = dict(N=N, corrector_steps=1, snr=0.5, eps=0.03, denoise=True, noise=torch.stack(draws).to(DEV))
    sep, nfe2 = eng.pc_sample(mix_norm, SDE, **kw)
    out = ops.scale_output(mix.to(DEV), sep)
    assert nfe == nfe2 == 60
    d, r = diff_rms(out, ref), rel_rms(out, ref)
    print(f"\n[split parity nf64 N30] out rms {rms(ref):.4f}  diff rms {d:.3e}  rel {r:.3e}")
    assert d < 1e-3 and r < 5e-4, f"waveform RMS difference {d:.3e} exceeds the 1e-3 bar"
    eng32, _ = engine(64, 2, _lib.F32)
    sep32, _ = eng32.pc_sample(mix_norm, SDE, **kw)
    s = si_sdr(sep, sep32)
    print(f"[split vs exact fp32 engine] rel rms {rel_rms(sep, sep32):.3e}  SI-SDR {s.flatten().tolist()}")
    assert float(s.min()) > 60.0


def test_split_graph_replay_and_determinism():
    eng, _ = engine(16, 2, _lib.F32_SPLIT)
    mix = torch.from_numpy(synth.synth_batch(3, T=6000)[0]).to(DEV)
    mn = ops.normalize_batch(mix)[0]
    a, _ = eng.pc_sample(mn, SDE, N=3, corrector_steps=1, seed=11)
    eng.set_graph(False)
    b, _ = eng.pc_sample(mn, SDE, N=3, corrector_steps=1, seed=11)
    eng.set_graph(True)
    assert torch.equal(a, b)


def test_split_engine_other_configurations_match_reference_golden(golden, golden2):
    # the remaining BASELINE configurations on the split engine: PriorMixSDE enhancement sampler, the other predictor /
    # corrector pairs, three sources, the published width nf = 128 (Cin = 512 concat inputs, 512-channel GroupNorm table)
    g, _ = golden
    eng, _ = engine(16, 2, _lib.F32_SPLIT)
    mix, draws, N, cs = _g9_inputs()
    mix_norm, _, _ = ops.normalize_batch(mix.to(DEV))
    psde = dict(kind=_lib.SDE_PRIORMIX, ndim=2, d_lambda=2.0, sigma_min=0.05, sigma_max=0.5, avg_len=510)
    sep, nfe = eng.pc_sample(mix_norm, psde, N=N, corrector_steps=cs, snr=0.5, eps=0.03, denoise=True, noise=draws.to(DEV))
    assert nfe == 6 and rel_rms(sep, g["g11_sep"]) < 3e-4 and diff_rms(sep, g["g11_sep"]) < 1e-3
    ts = O.scheduled_timesteps(N, 0.03, "log").numpy()
    a, _ = eng.pc_sample(mix_norm, SDE, N=N, corrector_steps=cs, snr=0.5, eps=0.03, predictor="euler_maruyama",
                         corrector="ald", noise=draws.to(DEV), timesteps=ts)
    assert rel_rms(a, g["g12_sep_em_ald_log"]) < 3e-4
    eng3, _ = engine(16, 3, _lib.F32_SPLIT)
    T = 4000
    xt3, mix3 = rnd("g7.xt3", (1, 3, T), 0.5), rnd("g7.mix3", (1, 1, T), 0.5)
    out3 = eng3.score(xt3.to(DEV), torch.tensor([0.3], device=DEV), mix3.to(DEV))
    assert rel_rms(out3, g["g7_score_S3"]) < 2e-4
    eng128, _ = engine(128, 2, _lib.F32_SPLIT, spec_factor=0.15)
    xt, mx = rnd("g7.xt", (1, 2, T), 0.5), rnd("g7.mix", (1, 1, T), 0.5)
    out = eng128.score(xt.to(DEV), torch.tensor([0.6], device=DEV), mx.to(DEV))
    assert rel_rms(out, golden2["g14_score_nf128"]) < 2e-4


def test_split_mixed_length_batch_equals_single_utterances():
    # per-utterance lengths and seeds (diffsep_pc_sample_ex) on the split engine: bit-for-bit like the fp32 engine
    eng, _ = engine(16, 2, _lib.F32_SPLIT)
    lens = [6000, 5200, 5999, 5100]
    assert len({eng.padded_frames(L) for L in lens}) == 1
    T = max(lens)
    mixn = torch.zeros(4, 1, T, device=DEV)
    for b, L in enumerate(lens):
        mixn[b, :, :L] = ops.normalize_batch(torch.from_numpy(synth.synth_mixture(b, T=L)[0])[None].to(DEV))[0][0]
    seeds = [5, 6, 7, 8]
    batch, _ = eng.pc_sample(mixn, SDE, N=2, corrector_steps=1, lengths=lens, seeds=seeds)
    for b, L in enumerate(lens):
        one, _ = eng.pc_sample(mixn[b:b + 1, :, :L].contiguous(), SDE, N=2, corrector_steps=1, seed=seeds[b])
        assert torch.equal(batch[b, :, :L], one[0]) and not bool(batch[b, :, L:].any())


@pytest.mark.parametrize("C1,C2,Cout,H,W", [(128, 0, 128, 16, 16), (128, 128, 128, 8, 8), (64, 192, 48, 16, 12), (256, 256, 256, 4, 1),
                                            (128, 64, 64, 8, 6), (128, 128, 128, 16, 24)])
@pytest.mark.parametrize("act,lazy", [(1, False), (1, True), (None, False)])
def test_split_small_image_conv3x3(C1, C2, Cout, H, W, act, lazy):
    # the small-image kernel (conv3x3_small.hip) on fp32 tensors with split products: the same cases as its bf16 test
    # (test_kernels_gpu.py::test_small_image_conv3x3), against torch fp32 on the CPU
    B, C = 3, C1 + C2
    tag = f"{C1}.{C2}.{H}.{W}"
    xa = (rnd("ssm.a" + tag, (B, H, W, C1), 1.2) + 0.1).to(DEV)
    xb = (rnd("ssm.b" + tag, (B, H, W, C2), 0.8) - 0.2).to(DEV) if C2 else None
    w = rnd(f"ssm.w{C}.{Cout}", (Cout, C, 3, 3), (9 * C) ** -0.5)
    bias, bb = rnd(f"ssm.bias{Cout}", (Cout,), 0.1).to(DEV), rnd(f"ssm.bb{Cout}", (B, Cout), 0.1).to(DEV)
    res = rnd(f"ssm.r{Cout}" + tag, (B, H, W, Cout)).to(DEV)
    groups = min(C // 4, 32)
    g, be = (1.0 + rnd(f"ssm.g{C}", (C,), 0.2)).to(DEV), rnd(f"ssm.be{C}", (C,), 0.1).to(DEV)
    xcat = torch.cat([xa, xb], -1).cpu() if C2 else xa.cpu()
    kw = {}
    if act is None:
        hn = xcat
    elif lazy:
        def acc(x):
            xd = x.double()
            return torch.stack([(xd.sum((1, 2)) * ops.STAT_SUM_SCALE).round(), ((xd * xd).sum((1, 2)) * ops.STAT_SQ_SCALE).round()],
                               -1).to(torch.int64).contiguous()
        kw = dict(gn_acc=(acc(xa), acc(xb) if C2 else None, g, be, groups), gn_act=act)
        hn = F.silu(F.group_norm(xcat.permute(0, 3, 1, 2), groups, g.cpu(), be.cpu(), eps=1e-6).permute(0, 2, 3, 1))
    else:
        sc, sh = (1.0 + rnd(f"ssm.sc{C}", (B, C), 0.2)).to(DEV), rnd(f"ssm.sh{C}", (B, C), 0.2).to(DEV)
        kw = dict(gn=(sc, sh), gn_act=act)
        hn = F.silu(xcat * sc.cpu()[:, None, None, :] + sh.cpu()[:, None, None, :])
    ref = F.conv2d(hn.permute(0, 3, 1, 2).double(), w.double(), bias.cpu().double(), padding=1).permute(0, 2, 3, 1)
    ref = ((ref + bb.cpu().double()[:, None, None, :] + res.cpu().double()) * 0.70710678).float()
    kc = ops.conv2d_chunk(3, torch.float32)
    for chunk in (0, kc):
        wp = ops.pack_conv_weight(w, torch.float32, chunk=chunk).to(DEV)
        y, st = ops.conv2d_fused(xa, wp, bias, Cout, 3, x2=xb, bias_b=bb, res=res, out_scale=0.70710678, stats=True,
                                 w_chunk=chunk, split=True, **kw)
        assert rel_rms(y.cpu(), ref) < 3e-5
        s = ops.stats_to_float(st).cpu()
        assert torch.allclose(s[..., 0], ref.double().sum((1, 2)), rtol=1e-4, atol=1e-4 * H * W)
        assert torch.allclose(s[..., 1], (ref.double() ** 2).sum((1, 2)), rtol=1e-4, atol=1e-4 * H * W)
