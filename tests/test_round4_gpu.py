"""Round-4 GPU tests, all through the C-ABI:
  * the time embedding in isolation against the reference golden and the oracle;
  * the graph cache of an engine is an LRU of bounded size (the Python API passes raw signal lengths);
  * the overflow net of the half-precision modes: an f16 run that overflows (one layer's weights scaled past 65504) is
    repeated on the split-precision twin by DiffSepModel itself — every caller gets it — and that result is what a split
    model returns directly and is inside the split engine's tolerance of the CPU oracle;
  * dtype="hybrid" outside the fused sampler (intermediate=True): the step-by-step loop keeps the head / tail schedule;
  * BASELINE configs[0] end to end: the separate CLI's written wav files against the CPU oracle on the device's own noise;
  * BASELINE configs[4] at nf = 64: three sources, N = 200 + 2 corrector steps = 600 evaluations, T = 100000, f16.
"""
import warnings

import numpy as np
import pytest
import torch

import diffsep_oracle as O
from diffsep_amd import _lib, ops, synth, wavio
from diffsep_amd.engine import Engine, pack_state_dict, param_table
from diffsep_amd.pl_model import DiffSepModel, default_config
from diffsep_amd import separate as sep_cli

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = "cuda"
SDE2 = dict(ndim=2, d_lambda=2.0, sigma_min=0.05, sigma_max=0.5)
SDE3 = dict(ndim=3, d_lambda=2.0, sigma_min=0.05, sigma_max=0.5)


def rms(a):
    return float(a.detach().double().pow(2).mean().sqrt())


def rel_rms(a, b):
    a = a.detach().double().cpu() if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a, np.float64))
    b = b.detach().double().cpu() if isinstance(b, torch.Tensor) else torch.as_tensor(np.asarray(b, np.float64))
    return rms(a - b) / (rms(b) + 1e-30)


def rnd(tag, shape, scale=1.0):
    return torch.from_numpy(synth.synth_noise(tag, shape)) * scale


def make_engine(nf, S, dtype, seed=7, **kw):
    cfg = _lib.model_config(nf=nf, num_sources=S, dtype=dtype)
    sd = synth.synth_state_dict([(n, s) for n, s, _ in param_table(cfg)], seed)
    return Engine(cfg, pack_state_dict(cfg, sd), **kw), sd


# ------------------------------------------------------------------------------------------------ a14: time embedding
@pytest.mark.parametrize("nf", [16, 64])
def test_time_embedding_matches_reference_golden(golden3, nf):
    cfg = O.default_config(nf, 2)
    sd = synth.synth_state_dict(O.param_table(cfg), 7)
    t = torch.tensor([1.0, 0.53, 0.2, 0.03])
    d = lambda k: torch.from_numpy(sd[k]).to(DEV)
    out = ops.time_embedding(t.to(DEV), d("all_modules.0.W"), d("all_modules.1.weight"), d("all_modules.1.bias"),
                             d("all_modules.2.weight"), d("all_modules.2.bias"))
    assert out.shape == (4, 4 * nf)
    assert rel_rms(out, golden3[f"g16_temb_nf{nf}"]) < 2e-5          # the reference's own modules (ncsnpp.py:324-343)
    assert rel_rms(out, O.time_embedding(O.to_torch(sd), t)) < 2e-5   # and the oracle


# ------------------------------------------------------------------------------------------------ graph cache
def test_graph_cache_is_a_bounded_lru():
    # a loop over utterances of distinct lengths through the Python API (DiffSepModel.separate passes the raw T): one
    # captured graph per (B, T) — the cache must stay bounded and a revisited, evicted plan must give the same samples
    eng, _ = make_engine(16, 2, _lib.F32)
    eng.set_option("graph_cache", 3)
    assert eng.get_option("graph_cache") == 3
    outs = {}
    lens = [4000, 4001, 5000, 12000, 12345, 20000, 4000, 12000]
    for T in lens:
        mix = torch.from_numpy(synth.synth_batch(1, T=T)[0]).to(DEV)
        mn, _, _ = ops.normalize_batch(mix)
        sep, nfe = eng.pc_sample(mn, SDE2, N=2, corrector_steps=1, snr=0.5, eps=0.03, denoise=True, seed=5)
        assert nfe == 4 and torch.isfinite(sep).all()
        assert eng.get_option("graphs_cached") <= 3
        if T in outs:
            assert torch.equal(outs[T], sep), f"T = {T}: an evicted and re-captured plan changed the samples"
        outs[T] = sep.clone()
    assert eng.get_option("graphs_cached") == 3
    with pytest.raises(_lib.DiffsepError):
        eng.set_option("no_such_option", 1)


# ------------------------------------------------------------------------------------------------ overflow -> split twin
def _overflowing_state(nf=16):
    """synthetic weights with ONE layer scaled so that its output (~1e5) is past the half-precision range (65504) but far
    inside fp32's: Conv_0 of the first residual block of the 4-row level (module 30; its 4 x W/64 pixels keep the GroupNorm
    accumulators of the following normalisation far from their own limit)"""
    cfg = _lib.model_config(nf=nf, num_sources=2)
    sd = synth.synth_state_dict([(n, s) for n, s, _ in param_table(cfg)], 7)
    sd["all_modules.30.Conv_0.weight"] = (sd["all_modules.30.Conv_0.weight"] * 2.0e5).astype(np.float32)
    return sd


def test_f16_overflow_is_repeated_on_the_split_twin_and_matches_the_oracle():
    sd = _overflowing_state()
    state = {"backbone." + k: torch.from_numpy(v) for k, v in sd.items()}
    B, T, N = 2, 4000, 2
    mix = torch.from_numpy(synth.synth_batch(B, T=T)[0]).to(DEV)
    m16 = DiffSepModel(default_config(nf=16), dtype="f16")
    m16.score_model.load_state_dict(state)
    (mn, _), *_ = m16.normalize_batch((mix, None))
    kw = dict(N=N, corrector_steps=1, snr=0.5, seed=11)
    # the f16 engine alone overflows: non-finite samples (check_finite=False = what the asynchronous CLIs call)
    raw, _ = m16.get_pc_sampler("reverse_diffusion", "ald2", mn, check_finite=False, **kw)()
    assert not bool(torch.isfinite(raw).all()), "the scaled layer was meant to overflow half precision"
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        est, nfe = m16.get_pc_sampler("reverse_diffusion", "ald2", mn, **kw)()
    assert nfe == 4 and bool(torch.isfinite(est).all())
    assert m16.fallback_batches == 1 and any("split-precision" in str(x.message) for x in w)
    # ... and what came back is exactly what a split model returns for the same request
    msp = DiffSepModel(default_config(nf=16), dtype="split")
    msp.score_model.load_state_dict(state)
    direct, _ = msp.get_pc_sampler("reverse_diffusion", "ald2", mn, **kw)()
    assert torch.equal(est, direct)
    assert msp.fallback_model() is None and msp.fallback_batches == 0
    # DiffSepModel.separate (the one-call API) goes through the same net
    assert bool(torch.isfinite(m16.separate(mix, N=N, seed=3)).all()) and m16.fallback_batches == 2
    # the split engine on these weights against the CPU oracle, injected noise: the parity the fallback restores
    draws = [rnd(f"ovf.z{i}", (B, 2, T)) for i in range(1 + 2 * N)]
    ref, _ = O.separate(O.to_torch(sd), O.default_config(16, 2), mix.cpu(), draws, N=N, corrector_steps=1, snr=0.5, eps=0.03)
    sep, _ = msp.engine().pc_sample(mn, SDE2, N=N, corrector_steps=1, snr=0.5, eps=0.03, noise=torch.stack(draws).to(DEV))
    out = ops.scale_output(mix, sep).cpu()
    assert rms(out - ref) < 1e-3 and rel_rms(out, ref) < 2e-3
    # a mode without a fallback raises instead of returning garbage: bf16 has the range, so force it through the net directly
    with pytest.raises(FloatingPointError):
        msp.rerun_if_nonfinite((raw, 4), lambda fb: (raw, 4))


def test_hybrid_model_outside_the_fused_sampler_keeps_its_schedule():
    # intermediate=True runs the step-by-step loop (round 3 raised for dtype="hybrid"): head_steps on the split model, the
    # rest on the f16 model.  With head_steps >= N the result is the split model's, with 0 the f16 model's.
    cfg = default_config(nf=16)
    sd = synth.synth_state_dict([(n, s) for n, s, _ in param_table(_lib.model_config(nf=16, num_sources=2))], 7)
    state = {"backbone." + k: torch.from_numpy(v) for k, v in sd.items()}
    B, T, N = 2, 4000, 3
    mix = torch.from_numpy(synth.synth_batch(B, T=T)[0]).to(DEV)

    def run(dtype, head_steps=None):
        m = DiffSepModel(cfg, dtype=dtype, head_steps=head_steps)
        m.score_model.load_state_dict(state)
        if m.tail_model is not None:
            m.tail_model.load_state_dict(state)
        (mn, _), *_ = m.normalize_batch((mix, None))
        torch.manual_seed(9)
        x, nfe, im = m.get_pc_sampler("reverse_diffusion", "ald2", mn, N=N, corrector_steps=1, snr=0.5, intermediate=True)()
        assert nfe == 2 * N and len(im) == N and torch.isfinite(x).all()
        return x

    full_head, split = run("hybrid", head_steps=N), run("split")
    assert rel_rms(full_head, split) < 1e-6        # (the split engines of the two library builds: the same fp32 code)
    no_head, f16 = run("hybrid", head_steps=0), run("f16")
    assert torch.equal(no_head, f16)
    mid = run("hybrid", head_steps=1)
    assert not torch.equal(mid, f16) and rel_rms(mid, split) < 2e-2


# ------------------------------------------------------------------------------------------------ configs[0] end to end
def test_configs0_separate_cli_wavs_match_the_oracle(tmp_path):
    # BASELINE configs[0]: separate.py on a folder of synthetic 8 kHz 2-speaker wavs, N = 30 in the reference; here N = 3 at
    # nf = 16 in fp32 so that the CPU oracle finishes in seconds.  The CLI draws its noise on the device: file i (sorted) gets the
    # i-th draw of a generator seeded with --seed as its RNG seed, draw d of the sampler is stream d of that seed — regenerated
    # here and handed to the oracle.
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    Ts = [4000, 4640, 4000, 5200]
    for i, T in enumerate(Ts):
        wavio.save(ind / f"utt{i}.wav", torch.from_numpy(synth.synth_mixture(i, T=T)[0]), 8000)
    N, seed = 3, 4
    sep_cli.main([str(ind), str(outd), "--synthetic-weights", "16", "-N", str(N), "--dtype", "f32", "--batch", "2", "--seed", str(seed)])
    seeds = torch.randint(0, 2 ** 62, (len(Ts),), generator=torch.Generator().manual_seed(seed)).tolist()
    cfg = O.default_config(16, 2)
    m = DiffSepModel(default_config(nf=16), dtype="f32")  # (--synthetic-weights: the model's own random init, seed 0)
    p = {k[len("backbone."):]: v.float() for k, v in m.score_model.state_dict().items()}
    for i, T in enumerate(Ts):
        mix, sr = wavio.load(ind / f"utt{i}.wav")  # (the 32-bit float file the CLI read)
        draws = [ops.randn_batch(1, 2, T, [seeds[i]], [T], d).cpu() for d in range(1 + 2 * N)]
        ref, nfe = O.separate(p, cfg, mix[None], draws, N=N, corrector_steps=1, snr=0.5, eps=0.03, denoise=True)
        got = torch.stack([wavio.load(outd / f"s{k}" / f"utt{i}.wav")[0][0] for k in range(2)])
        assert sr == 8000 and got.shape == (2, T) and nfe == 2 * N
        assert rel_rms(got, ref[0]) < 1e-4, f"utt{i}: written wav vs oracle {rel_rms(got, ref[0]):.2e}"


# ------------------------------------------------------------------------------------------------ configs[4] at nf = 64
def test_configs4_nf64_three_sources_600_evaluations_f16():
    eng, sd = make_engine(64, 3, _lib.F16)
    T, N, cs = 100000, 200, 2
    assert eng.padded_frames(T) == 832
    mix = torch.from_numpy(synth.synth_mixture(1, T=T, n_src=3)[0])[None]
    # one score evaluation of the 3-source nf = 64 network at this length against the CPU oracle (f16 storage: 5e-3)
    cfg = O.default_config(64, 3)
    mixn, _, _ = O.normalize_batch(mix)
    xt = O.prior_sampling(cfg, mixn, rnd("c4n64.z", (1, 3, T)))
    t = torch.tensor([0.31])
    ref = O.score_forward(O.to_torch(sd), cfg, xt, t, mixn)
    r = rel_rms(eng.score(xt.to(DEV), t.to(DEV), mixn.to(DEV)), ref)
    print(f"\n[configs[4] nf64 S3 T=100000, one score evaluation, f16 vs oracle] rel rms {r:.3e}")
    assert r < 5e-3
    mn = mixn.to(DEV)
    kw = dict(N=N, corrector_steps=cs, snr=0.5, eps=0.03, denoise=True, seed=21)
    out, nfe = eng.pc_sample(mn, SDE3, **kw)
    assert nfe == N * (1 + cs) == 600 and out.shape == (1, 3, T)
    assert torch.isfinite(out).all() and 1e-3 < rms(out) < 1e3
    again, _ = eng.pc_sample(mn, SDE3, **kw)
    assert torch.equal(again, out)                 # deterministic
    eng.set_graph(False)
    eager, _ = eng.pc_sample(mn, SDE3, **kw)
    eng.set_graph(True)
    assert torch.equal(eager, out)                 # graph replay == eager launches


# ------------------------------------------------------------------------------------------------ fused attention block
@pytest.mark.parametrize("kind,dt,tol", [("f16", torch.float16, 4e-3), ("bf16", torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("B,H,W", [(3, 16, 16), (2, 4, 4), (2, 16, 8), (1, 16, 1)])
def test_fused_attention_block_matches_the_oracle(kind, dt, tol, B, H, W):
    # AttnBlockpp with 128 channels on <= 256 pixels: ONE kernel in the 16-bit engines (attn_fused.hip; the key projection folded
    # into the query side, K never formed) against the CPU oracle's block (layerspp.py:76-92), and against the unfused launch
    # sequence of rounds 1 - 3 on the same operands (option no_attn_fused)
    C = 128
    tbl = [("GroupNorm_0.weight", (C,)), ("GroupNorm_0.bias", (C,))]
    for i in range(4):
        tbl += [(f"NIN_{i}.W", (C, C)), (f"NIN_{i}.b", (C,))]
    sd = synth.synth_state_dict(tbl, 5)
    for i in range(4):  # (the synthetic NIN scale is tiny: make the attention logits and the output matter)
        sd[f"NIN_{i}.W"] = (sd[f"NIN_{i}.W"] * 3.0).astype(np.float32)
        sd[f"NIN_{i}.b"] = (sd[f"NIN_{i}.b"] + 0.1 * synth.synth_noise(f"fa.b{i}", (C,))).astype(np.float32)
    x = rnd(f"fa.x{B}{H}{W}", (B, C, H, W), 1.3) + 0.2
    xq = x.to(dt).float()  # the kernel sees the rounded input
    ref = O._attn_block(O.to_torch(sd), "", xq)
    params = [sd[n] for n, _ in tbl]
    xd = ops.to_nhwc(x).to(dt).to(DEV)
    lib = _lib.lib(kind)
    y = ops.attnblock_forward(params, xd)
    r = rel_rms(ops.to_nchw(y).float(), ref)
    _lib.check(lib.diffsep_set_option(b"no_attn_fused", 1), lib)
    try:
        y0 = ops.attnblock_forward(params, xd)
    finally:
        _lib.check(lib.diffsep_set_option(b"no_attn_fused", 0), lib)
    r0 = rel_rms(ops.to_nchw(y0).float(), ref)
    print(f"\n[attention {kind} B{B} {H}x{W}] fused {r:.2e}, unfused {r0:.2e} rel rms vs oracle")
    assert r < tol and r0 < tol
    assert not torch.equal(y, y0) or H * W <= 16  # (two different launch sequences: the option does switch)


# ------------------------------------------------------------------------------------------------ nf = 128: split cat(128, 128) blocks
def test_nf128_cat_blocks_as_two_register_weight_launches_vs_oracle():
    # the published width at full size: the cat(128, 128) -> 128 residual blocks of the 256^2 / 128^2 levels.  Round 4 ran each
    # convolution as two 128-channel register-weight launches chained through the residual (engine.hip res_block); since round 5
    # the streamed-weight kernel takes them whole (option no_sw restores the two-launch route, no_sw + no_split256 the one-launch
    # generic tile of rounds 1 - 3).  All three against the CPU oracle on the same operands.
    cfg = O.default_config(128, 2, spec_factor=0.15)
    mcfg = _lib.model_config(nf=128, num_sources=2, dtype=_lib.F16, spec_factor=0.15)
    sd = synth.synth_state_dict([(n, s) for n, s, _ in param_table(mcfg)], 7)
    eng = Engine(mcfg, pack_state_dict(mcfg, sd))
    T = 32000
    mix = torch.from_numpy(synth.synth_batch(1, T=T)[0])
    mixn, _, _ = O.normalize_batch(mix)
    xt = O.prior_sampling(cfg, mixn, rnd("s256.z", (1, 2, T)))
    t = torch.tensor([0.4])
    ref = O.score_forward(O.to_torch(sd), cfg, xt, t, mixn)
    c = eng.score(xt.to(DEV), t.to(DEV), mixn.to(DEV))
    eng.set_option("no_sw", 1)
    a = eng.score(xt.to(DEV), t.to(DEV), mixn.to(DEV))
    eng.set_option("no_split256", 1)
    b = eng.score(xt.to(DEV), t.to(DEV), mixn.to(DEV))
    eng.set_option("no_split256", 0)
    eng.set_option("no_sw", 0)
    ra, rb, rc = rel_rms(a, ref), rel_rms(b, ref), rel_rms(c, ref)
    print(f"\n[nf128 f16 score, T=32000 vs oracle] streamed weights {rc:.3e}, two launches per convolution {ra:.3e}, one launch {rb:.3e}")
    assert ra < 8e-3 and rb < 8e-3 and rc < 8e-3 and not torch.equal(a, b) and not torch.equal(a, c)
