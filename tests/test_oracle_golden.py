"""Pin the CPU oracle (oracle/diffsep_oracle.py) against outputs of the reference itself
(tests/golden/golden_ref.npz, produced by tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import diffsep_oracle as O
from diffsep_amd import synth

torch.set_grad_enabled(False)


def rel_rms(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / (np.sqrt(np.mean(b ** 2)) + 1e-30))


def weights(cfg, seed):
    return O.to_torch(synth.synth_state_dict(O.param_table(cfg), seed))


def test_param_table_matches_reference_state_dict(golden):
    _, meta = golden
    for key, cfg in (("param_table_nf16_S2", O.default_config(16, 2)), ("param_table_nf16_S3", O.default_config(16, 3)),
                     ("param_table_nf64_S2", O.default_config(64, 2)), ("param_table_nf128_S2", O.default_config(128, 2))):
        mine = [[n, list(s)] for n, s in O.param_table(cfg)]
        assert mine == meta[key], key
    n64 = sum(int(np.prod(s)) for _, s in O.param_table(O.default_config(64, 2)))
    assert n64 == meta["param_count_nf64_S2"] == 16456390  # SURVEY.md §5 / BASELINE.md
    assert meta["param_count_nf128_S2"] == 65623366


def test_fir_resampling(golden):
    g, _ = golden
    x = torch.from_numpy(synth.synth_noise("g1.x", (2, 8, 6, 10)))
    assert np.abs(O.fir_up2(x).numpy() - g["g1_up"]).max() < 1e-6
    assert np.abs(O.fir_down2(x).numpy() - g["g1_down"]).max() < 1e-6


def _block_weights(names_shapes, seed):
    return O.to_torch(synth.synth_state_dict(names_shapes, seed))


def test_resblocks_and_attention(golden):
    g, _ = golden
    temb = torch.from_numpy(synth.synth_noise("g4.temb", (2, 32)))
    for tag, cout, up, down in (("plain", 16, False, False), ("widen", 24, False, False), ("up", 16, True, False),
                                ("down", 16, False, True)):
        cin = 16
        tbl = [("GroupNorm_0.weight", (cin,)), ("GroupNorm_0.bias", (cin,)), ("Conv_0.weight", (cout, cin, 3, 3)),
               ("Conv_0.bias", (cout,)), ("Dense_0.weight", (cout, 32)), ("Dense_0.bias", (cout,)),
               ("GroupNorm_1.weight", (cout,)), ("GroupNorm_1.bias", (cout,)), ("Conv_1.weight", (cout, cout, 3, 3)),
               ("Conv_1.bias", (cout,))]
        if cin != cout or up or down:
            tbl += [("Conv_2.weight", (cout, cin, 1, 1)), ("Conv_2.bias", (cout,))]
        p = _block_weights(tbl, 4)
        x = torch.from_numpy(synth.synth_noise("g4.x." + tag, (2, cin, 8, 12)))
        y = O._res_block(p, "", x, temb, up=up, down=down)
        assert rel_rms(y.numpy(), g["g4_" + tag]) < 2e-6, tag
    for tag, hw in (("16x16", (16, 16)), ("4x4", (4, 4))):
        tbl = [("GroupNorm_0.weight", (16,)), ("GroupNorm_0.bias", (16,))]
        for k in range(4):
            tbl += [(f"NIN_{k}.W", (16, 16)), (f"NIN_{k}.b", (16,))]
        p = _block_weights(tbl, 5)
        x = torch.from_numpy(synth.synth_noise("g5.x." + tag, (2, 16) + hw))
        assert rel_rms(O._attn_block(p, "", x).numpy(), g["g5_" + tag]) < 2e-6, tag


def test_frame_counts_are_exact(golden):
    _, meta = golden
    cfg = O.default_config(16, 2)
    for T, (F, W) in meta["frames"].items():
        T = int(T)
        assert O.num_frames(cfg, T) == F
        assert 64 * ((F + 63) // 64) == W
    assert meta["frames"]["32000"] == [253, 256] and meta["frames"]["4000"] == [35, 64]


def test_pre_post_process(golden):
    g, _ = golden
    cfg = O.default_config(16, 2)
    x = torch.from_numpy(synth.synth_noise("g6.x.4000", (1, 3, 4000))) * 0.3
    spec, T, n_pad = O.pre_process(cfg, x)
    assert spec.shape == (1, 6, 256, 64) and n_pad == 29
    assert rel_rms(spec.numpy(), g["g6_pre_4000"]) < 1e-6
    y = torch.from_numpy(synth.synth_noise("g6.y.4000", (1, 4, 256, 64))) * 0.2
    assert rel_rms(O.post_process(cfg, y, T, n_pad).numpy(), g["g6_post_4000"]) < 1e-6
    for T in (31999, 32000, 32001):
        x = torch.from_numpy(synth.synth_noise(f"g6.x.{T}", (1, 3, T))) * 0.3
        spec, _, n_pad = O.pre_process(cfg, x)
        F = spec.shape[-1] - n_pad
        assert rel_rms(spec[..., [0, 1, F // 2, F - 2, F - 1]].numpy(), g[f"g6_pre_{T}_frames"]) < 1e-6


def test_score_model_forward(golden):
    g, _ = golden
    cfg = O.default_config(16, 2)
    p = weights(cfg, 7)
    T = 4000
    xt = torch.from_numpy(synth.synth_noise("g7.xt", (2, 2, T))) * 0.5
    mix = torch.from_numpy(synth.synth_noise("g7.mix", (2, 1, T))) * 0.5
    t = torch.tensor([0.7, 0.05])
    assert rel_rms(O.score_forward(p, cfg, xt, t, mix).numpy(), g["g7_score"]) < 2e-5
    xb = torch.from_numpy(synth.synth_noise("g7.xb", (1, 6, 256, 64))) * 0.3
    assert rel_rms(O.ncsnpp_forward(p, cfg, xb, torch.tensor([0.4])).numpy(), g["g7_backbone"]) < 2e-5
    cfg3 = O.default_config(16, 3)
    p3 = weights(cfg3, 7)
    xt3 = torch.from_numpy(synth.synth_noise("g7.xt3", (1, 3, T))) * 0.5
    mix3 = torch.from_numpy(synth.synth_noise("g7.mix3", (1, 1, T))) * 0.5
    assert rel_rms(O.score_forward(p3, cfg3, xt3, torch.tensor([0.3]), mix3).numpy(), g["g7_score_S3"]) < 2e-5


def test_sde_tables(golden):
    g, _ = golden
    cfg = O.default_config()
    ts = torch.linspace(1.0, 0.03, 30)
    assert np.array_equal(ts.numpy(), g["g8_timesteps"])
    ev1, ev2 = O.cov_eigval(cfg, ts)
    np.testing.assert_allclose(ev1.numpy(), g["g8_ev1"], rtol=2e-6)
    np.testing.assert_allclose(ev2.numpy(), g["g8_ev2"], rtol=2e-6)
    np.testing.assert_allclose(O.mix_std(cfg, ts, 2).numpy(), g["g8_std"], rtol=2e-6, atol=1e-9)
    # SURVEY.md Appendix B known answers (float64 probe of the reference)
    assert abs(float(ev1[0]) - 2.475e-01) < 1e-6 and abs(float(ev2[0]) - 1.337663e-01) < 1e-6
    L1 = O.mix_std(cfg, torch.tensor([1.0]), 2)[0]
    assert abs(float(L1[0, 0]) - 0.4316172) < 1e-6 and abs(float(L1[0, 1]) - 0.0658765) < 1e-6


def test_pc_sampler_and_separate(golden):
    g, meta = golden
    cfg = O.default_config(16, 2)
    p = weights(cfg, 7)
    B, S, T, N, cs = 2, 2, 4000, 3, 1
    mix = torch.from_numpy(synth.synth_batch(B, T=T)[0])
    mix_norm, _, _ = O.normalize_batch(mix)
    assert rel_rms(mix_norm.numpy(), g["g10_mix_norm"]) < 1e-6
    draws = [torch.from_numpy(synth.synth_noise(f"g9.z{i}", (B, S, T))) for i in range(1 + N * (cs + 1))]
    assert rel_rms(O.prior_sampling(cfg, mix_norm, draws[0]).numpy(), g["g9_prior"]) < 1e-6
    sep, nfe = O.pc_sampler(p, cfg, mix_norm, draws, N=N, corrector_steps=cs, snr=0.5, eps=0.03, denoise=True)
    assert nfe == meta["g9_nfe"] == 6
    assert rel_rms(sep.numpy(), g["g9_sep"]) < 1e-4
    sep2, _ = O.pc_sampler(p, cfg, mix_norm, draws, N=N, corrector_steps=cs, snr=0.5, eps=0.03, denoise=False)
    assert rel_rms(sep2.numpy(), g["g9_sep_nodenoise"]) < 1e-4
    # isolated updates
    x0 = torch.from_numpy(synth.synth_noise("g9.x0", (B, S, T))) * 0.5
    tv = torch.tensor([0.8, 0.2])
    sc = O.score_forward(p, cfg, x0, tv, mix_norm)
    xc, xcm = O.corrector_ald2(cfg, x0, tv, sc, draws[1], 0.5)
    xp, xpm = O.predictor_reverse_diffusion(cfg, x0, tv, sc, draws[2], N)
    for a, k in ((xc, "g9_corr_x"), (xcm, "g9_corr_mean"), (xp, "g9_pred_x"), (xpm, "g9_pred_mean")):
        assert rel_rms(a.numpy(), g[k]) < 2e-5, k
    # separate.separate() on utterance 0 (N=2) and scale_output
    d1 = [torch.from_numpy(synth.synth_noise(f"g10.z{i}", (1, S, T))) for i in range(5)]
    out, _ = O.separate(p, cfg, mix[:1], d1, N=2, corrector_steps=1, snr=0.5, eps=0.03, denoise=True)
    assert rel_rms(out.numpy(), g["g10_separate"]) < 1e-4
    assert rel_rms(O.scale_output(mix, torch.from_numpy(g["g9_sep"])).numpy(), g["g10_scale_output"]) < 1e-6


def test_priormix_sde_enhancement_path(golden):
    """PriorMixSDE (config/model/nr.yaml, evaluate.py --enhance): sigma_mix, prior, isolated updates, full sampler."""
    g, _ = golden
    cfg = O.default_config(16, 2)
    p = weights(cfg, 7)
    B, S, T, N, cs = 2, 2, 4000, 3, 1
    mix = torch.from_numpy(synth.synth_batch(B, T=T)[0])
    mix_norm, _, _ = O.normalize_batch(mix)
    smix = O.sigma_mix(mix_norm, 510)
    assert smix.shape == (B, 1, T) and rel_rms(smix.numpy(), g["g11_sigma_mix"]) < 1e-6
    draws = [torch.from_numpy(synth.synth_noise(f"g9.z{i}", (B, S, T))) for i in range(1 + N * (cs + 1))]
    assert rel_rms(O.prior_sampling(cfg, mix_norm, draws[0], smix).numpy(), g["g11_prior"]) < 1e-6
    x0 = torch.from_numpy(synth.synth_noise("g9.x0", (B, S, T))) * 0.5
    tv = torch.tensor([0.8, 0.2])
    sc = O.score_forward(p, cfg, x0, tv, mix_norm)
    xc, xcm = O.corrector_ald2(cfg, x0, tv, sc, draws[1], 0.5, smix)
    xp, xpm = O.predictor_reverse_diffusion(cfg, x0, tv, sc, draws[2], N, smix)
    for a, k in ((xc, "g11_corr_x"), (xcm, "g11_corr_mean"), (xp, "g11_pred_x"), (xpm, "g11_pred_mean")):
        assert rel_rms(a.numpy(), g[k]) < 2e-5, k
    sep, nfe = O.pc_sampler(p, cfg, mix_norm, draws, N=N, corrector_steps=cs, snr=0.5, eps=0.03, denoise=True,
                            priormix_avg_len=510)
    assert nfe == 6 and rel_rms(sep.numpy(), g["g11_sep"]) < 1e-4


def test_remaining_sampler_surface(golden):
    """euler_maruyama == reverse_diffusion, 'ald' and 'langevin' correctors, scheduled sampler."""
    g, meta = golden
    cfg = O.default_config(16, 2)
    p = weights(cfg, 7)
    B, S, T, N, cs = 2, 2, 4000, 3, 1
    mix_norm = torch.from_numpy(g["g10_mix_norm"])
    draws = [torch.from_numpy(synth.synth_noise(f"g9.z{i}", (B, S, T))) for i in range(7)]
    x0 = torch.from_numpy(synth.synth_noise("g9.x0", (B, S, T))) * 0.5
    tv = torch.tensor([0.8, 0.2])
    sc = O.score_forward(p, cfg, x0, tv, mix_norm)
    xe, xem = O.predictor_reverse_diffusion(cfg, x0, tv, sc, draws[2], N)  # the same update as euler_maruyama
    assert rel_rms(xe.numpy(), g["g12_em_x"]) < 2e-5 and rel_rms(xem.numpy(), g["g12_em_mean"]) < 2e-5
    xa, xam = O.corrector_ald(cfg, x0, tv, sc, draws[1], 0.5)
    assert rel_rms(xa.numpy(), g["g12_ald_x"]) < 2e-5 and rel_rms(xam.numpy(), g["g12_ald_mean"]) < 2e-5
    xl, xlm = O.corrector_langevin(x0, sc, draws[1], 0.5)
    assert rel_rms(xl.numpy(), g["g12_langevin_x"]) < 2e-5 and rel_rms(xlm.numpy(), g["g12_langevin_mean"]) < 2e-5
    assert meta["g12_pflow_G_is_zero"]
    a, _ = O.pc_sampler(p, cfg, mix_norm, draws, N=N, corrector_steps=cs, snr=0.5, eps=0.03, corrector="ald",
                        timesteps=O.scheduled_timesteps(N, 0.03, "log"))
    assert rel_rms(a.numpy(), g["g12_sep_em_ald_log"]) < 1e-4
    b, _ = O.pc_sampler(p, cfg, mix_norm, draws, N=N, corrector_steps=cs, snr=0.5, eps=0.03, corrector="langevin")
    assert rel_rms(b.numpy(), g["g12_sep_rd_langevin"]) < 1e-4


def test_sde_object_surface(golden, golden2):
    """sde / marginal_prob / mult_std / discretize / reverse().discretize of MixSDE and PriorMixSDE
    (sdes/sdes.py:93-173,275-328,451-537) against the reference's own outputs."""
    g, g2 = golden[0], golden2
    cfg = O.default_config(16, 2)
    p = weights(cfg, 7)
    B, S, T, N = 2, 2, 4000, 3
    mix_norm = torch.from_numpy(g["g10_mix_norm"])
    x0 = torch.from_numpy(synth.synth_noise("g9.x0", (B, S, T))) * 0.5
    tv = torch.tensor([0.8, 0.2])
    sc = O.score_forward(p, cfg, x0, tv, mix_norm)
    for tag, smix in (("mix", None), ("pmix", O.sigma_mix(mix_norm, 510)[:, 0])):
        drift, diff = O.sde_coefficients(cfg, x0, tv, smix)
        assert rel_rms(drift.numpy(), g2[f"g13_{tag}_drift"]) < 1e-6 and rel_rms(diff.numpy(), g2[f"g13_{tag}_diffusion"]) < 1e-6
        assert diff.shape == g2[f"g13_{tag}_diffusion"].shape
        assert rel_rms(O.sde_mean(cfg, x0, tv).numpy(), g2[f"g13_{tag}_mean"]) < 1e-6
        std = O.sde_std(cfg, tv, S, smix)
        assert std.shape == g2[f"g13_{tag}_std"].shape and rel_rms(std.numpy(), g2[f"g13_{tag}_std"]) < 1e-6
        assert rel_rms(O.sde_mult_std(std, x0).numpy(), g2[f"g13_{tag}_mult_std"]) < 1e-6
        f, G = O.sde_discretize(cfg, x0, tv, N, smix)
        assert rel_rms(f.numpy(), g2[f"g13_{tag}_f"]) < 1e-6 and rel_rms(G.numpy(), g2[f"g13_{tag}_G"]) < 1e-6
        rf, rG = O.rsde_discretize(cfg, x0, tv, sc, N, smix)
        assert rel_rms(rf.numpy(), g2[f"g13_{tag}_rev_f"]) < 2e-5 and rel_rms(rG.numpy(), g2[f"g13_{tag}_rev_G"]) < 1e-6
        dp = diff if diff.dim() == 3 else diff[:, None, None]
        assert rel_rms((drift - dp ** 2 * sc).numpy(), g2[f"g13_{tag}_rsde_drift"]) < 2e-5


def test_published_width_nf128(golden, golden2):
    """nf = 128, spec_factor 0.15 (icassp-separation.yaml:14-18, nr.yaml): one score evaluation and the PriorMixSDE
    sampler against the reference."""
    g, g2 = golden[0], golden2
    cfg = O.default_config(128, 2, spec_factor=0.15)
    p = weights(cfg, 7)
    S, T = 2, 4000
    xt = torch.from_numpy(synth.synth_noise("g7.xt", (1, S, T))) * 0.5
    mx = torch.from_numpy(synth.synth_noise("g7.mix", (1, 1, T))) * 0.5
    assert rel_rms(O.score_forward(p, cfg, xt, torch.tensor([0.6]), mx).numpy(), g2["g14_score_nf128"]) < 2e-5
    mix_norm = torch.from_numpy(g["g10_mix_norm"])[:1]
    draws = [torch.from_numpy(synth.synth_noise(f"g14.z{i}", (1, S, T))) for i in range(5)]
    sep, nfe = O.pc_sampler(p, cfg, mix_norm, draws, N=2, corrector_steps=1, snr=0.5, eps=0.03, denoise=True,
                            priormix_avg_len=510)
    assert nfe == 4 and rel_rms(sep.numpy(), g2["g14_priormix_sep_nf128"]) < 1e-4


def test_three_source_updates(golden, golden2):
    """S = 3: std matrix, ald2 corrector and reverse-diffusion predictor updates against the reference (its
    MixSDE.prior_sampling is undefined for S = 3 — quirk Q2 — so the full 3-source sampler has no reference vector)."""
    g, g2 = golden[0], golden2
    cfg = O.default_config(16, 3)
    p = weights(cfg, 7)
    B, T, N = 2, 4000, 3
    mix_norm = torch.from_numpy(g["g10_mix_norm"])
    x0 = torch.from_numpy(synth.synth_noise("g15.x0", (B, 3, T))) * 0.5
    z = [torch.from_numpy(synth.synth_noise(f"g15.z{i}", (B, 3, T))) for i in range(2)]
    tv = torch.tensor([0.8, 0.2])
    assert rel_rms(O.mix_std(cfg, tv, 3).numpy(), g2["g15_std"]) < 1e-6
    sc = O.score_forward(p, cfg, x0, tv, mix_norm)
    xc, xcm = O.corrector_ald2(cfg, x0, tv, sc, z[0], 0.5)
    xp, xpm = O.predictor_reverse_diffusion(cfg, x0, tv, sc, z[1], N)
    for a, k in ((xc, "g15_corr_x"), (xcm, "g15_corr_mean"), (xp, "g15_pred_x"), (xpm, "g15_pred_mean")):
        assert rel_rms(a.numpy(), g2[k]) < 2e-5, k


def test_separation_metrics_restatement():
    """oracle.si_bss_eval_sources (parity unpinned: fast_bss_eval is absent) against closed forms and its own
    definitions in the time domain, incl. the degenerate cases."""
    rng = np.random.default_rng(3)
    T = 8000
    r = rng.standard_normal((1, 3, T))
    # estimates = permuted references + orthogonal noise of relative power 1e-2 -> SI-SDR = 20 dB each
    n = rng.standard_normal((1, 3, T))
    for k in range(3):
        for j in range(3):
            n[0, k] -= (n[0, k] @ r[0, j]) / (r[0, j] @ r[0, j]) * r[0, j]
        n[0, k] *= 0.1 * np.linalg.norm(r[0, k]) / np.linalg.norm(n[0, k])
    # (the references are only approximately orthogonal to each other: the projection on r_k is taken exactly)
    est = (r + n)[:, [2, 0, 1]]
    sdr, sir, sar, perm = O.si_bss_eval_sources(r, est)
    assert perm.tolist() == [[1, 2, 0]]
    assert np.allclose(sar, 20.0, atol=0.05) and (sir > 25).all() and (np.abs(sdr - 20.0) < 0.3).all()
    # identical estimate: everything saturates at +clamp_db
    sdr, sir, sar, _ = O.si_bss_eval_sources(r, r.copy(), clamp_db=100.0)
    assert np.allclose(sdr, 100.0) and np.allclose(sar, 100.0)
    # a silent reference / a silent estimate: clamped at -clamp_db, no NaN
    r0 = r.copy(); r0[0, 1] = 0.0
    out = O.si_bss_eval_sources(r0, est)
    assert all(np.isfinite(v).all() for v in out[:3]) and out[0].min() == -100.0
    e0 = est.copy(); e0[0, 0] = 0.0
    out = O.si_bss_eval_sources(r, e0)
    assert all(np.isfinite(v).all() for v in out[:3]) and out[0].min() == -100.0
    # two identical references (singular Gram matrix): finite
    r2 = r.copy(); r2[0, 2] = r2[0, 1]
    assert all(np.isfinite(v).all() for v in O.si_bss_eval_sources(r2, est)[:3])


@pytest.mark.parametrize("nf", [16, 64])
def test_time_embedding_in_isolation(golden3, nf):
    # GaussianFourierProjection(log t) -> Linear -> SiLU -> Linear (ncsnpp.py:324-343) against the
    # reference backbone's own modules
    cfg = O.default_config(nf, 2)
    p = O.to_torch(synth.synth_state_dict(O.param_table(cfg), 7))
    t = torch.tensor([1.0, 0.53, 0.2, 0.03])
    out = O.time_embedding(p, t)
    assert out.shape == (4, 4 * nf)
    assert rel_rms(out, golden3[f"g16_temb_nf{nf}"]) < 2e-5
