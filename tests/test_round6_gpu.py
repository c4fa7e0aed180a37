"""Round 6:

* the SHIPPED dispatch against the oracle directly: the oracle's full-size utterance rides in row 0 of a B = 16 f16 batch (the
  register-weight and streamed-weight kernels run at every level there, unlike at B = 1) on its 61 injected draws;
  the same for the published width in the mode dtype="auto" ships there (nf = 128, hybrid) at B = 2;
* the throughput mode the multi-stream callers set (engine option rw_quarter) changes where blocks run and how the fp32 partial
  sums of the GroupNorm statistics are grouped, nothing else: agreement at the rounding level of the 16-bit mode (what a
  different batch size also costs), no effect at all on the fp32 / split engines;
* the reference's evaluate.py command line runs verbatim on an experiment folder laid out like the reference's, and writes
  the reference's output tree;
* per-class profile: the streamed-weight kernels have classes of their own (they were summed into the attention class).
"""
import json

import numpy as np
import pytest
import torch

import diffsep_oracle as O
from diffsep_amd import _lib, ops, synth, wavio
from diffsep_amd.engine import Engine, pack_state_dict, param_table

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = "cuda"
SDE = dict(ndim=2, d_lambda=2.0, sigma_min=0.05, sigma_max=0.5)


def _rms(a):
    return float(a.double().pow(2).mean().sqrt())


def _engine(nf, dtype, seed=7, spec_factor=0.33, lib_kind=None):
    cfg = _lib.model_config(nf=nf, num_sources=2, dtype=dtype, spec_factor=spec_factor)
    sd = synth.synth_state_dict([(n, s) for n, s, _ in param_table(cfg)], seed)
    return Engine(cfg, pack_state_dict(cfg, sd), lib_kind=lib_kind), sd


def test_shipped_f16_dispatch_at_batch_16_against_the_oracle(oracle_fullsize_nf64):
    fs = oracle_fullsize_nf64
    T, N, ref = fs["T"], fs["N"], fs["ref"]
    B = 16
    eng, _ = _engine(64, _lib.F16)
    mix = torch.from_numpy(synth.synth_batch(B, T=T)[0])
    assert torch.equal(mix[:1], fs["mix"])  # (row 0 IS the oracle's utterance; rows 1 - 15 are other mixtures)
    mix = mix.to(DEV)
    mix_norm, _, _ = ops.normalize_batch(mix)
    noise = torch.randn((1 + 2 * N, B, 2, T), generator=torch.Generator().manual_seed(11)).to(DEV)
    noise[:, 0] = torch.stack(fs["draws"])[:, 0].to(DEV)
    sep, nfe = eng.pc_sample(mix_norm, SDE, N=N, corrector_steps=1, snr=0.5, eps=0.03, denoise=True, noise=noise)
    assert nfe == 60 and torch.isfinite(sep).all()
    out0 = ops.scale_output(mix[:1], sep[:1].contiguous()).cpu()
    d = _rms(out0 - ref)
    r = d / _rms(ref)
    print(f"\n[B=16 f16, row 0 vs the CPU oracle, injected noise] diff rms {d:.3e}  rel {r:.3e}")
    assert d < 1e-3 and r < 1e-2, f"shipped f16 dispatch at B = 16: {d:.3e} abs / {r:.3e} rel RMS from the oracle"
    # ... and that WAS the shipped dispatch: the kernels of one evaluation of the same batch shape
    eng.profile_begin()
    eng.pc_sample(mix_norm, SDE, N=1, corrector_steps=1, snr=0.5, eps=0.03, denoise=True, seed=3)
    prof = eng.profile_end()
    recs = eng.profile_records()
    names = {r_["kernel"].split("<")[0] for r_ in recs}
    assert "conv3x3_rw_kernel" in names and "conv3x3_sw_kernel" in names, names
    # streamed-weight launches have their own class (10), attention keeps 9
    assert all(r_["cls"] == 10 for r_ in recs if r_["kernel"].startswith("conv3x3_sw_kernel"))
    assert all(r_["cls"] == 9 for r_ in recs if r_["kernel"].startswith("attn_fused"))
    assert prof["conv3x3_sw_streamed"][2] > 0 and prof["attention_fused"][2] > 0
    assert prof["attention_fused"][2] == sum(1 for r_ in recs if r_["kernel"].startswith("attn_fused"))
    assert prof["attention_fused"][1] < prof["conv3x3_sw_streamed"][1]
    eng.close()


def test_published_width_hybrid_at_batch_2_against_the_oracle():
    # nf = 128, spec_factor 0.15 (icassp-separation.yaml:14-18), 1 s of audio, N = 30 + 1 corrector step; dtype="auto" ships
    # "hybrid" at this width: a split-precision engine for the first 5 reverse steps, the f16 engine after
    from diffsep_amd.pl_model import HYBRID_HEAD_STEPS
    nf, T, N, B, SF = 128, 8000, 30, 2, 0.15
    cfg = O.default_config(nf, 2, spec_factor=SF)
    e16, sd = _engine(nf, _lib.F16, spec_factor=SF)
    esp, _ = _engine(nf, _lib.F32_SPLIT, spec_factor=SF, lib_kind="f16")
    mix = torch.from_numpy(synth.synth_batch(B, T=T)[0])
    draws = [torch.from_numpy(synth.synth_noise(f"r6.z{i}", (1, 2, T))) for i in range(1 + 2 * N)]
    ref, nfe = O.separate(O.to_torch(sd), cfg, mix[:1], draws, N=N, corrector_steps=1, snr=0.5, eps=0.03, denoise=True)
    mix_d = mix.to(DEV)
    mix_norm, _, _ = ops.normalize_batch(mix_d)
    noise = torch.randn((1 + 2 * N, B, 2, T), generator=torch.Generator().manual_seed(12)).to(DEV)
    noise[:, 0] = torch.stack(draws)[:, 0].to(DEV)
    sep, nfe2 = e16.pc_sample(mix_norm, SDE, N=N, corrector_steps=1, snr=0.5, eps=0.03, denoise=True, noise=noise,
                              tail=esp, head_steps=HYBRID_HEAD_STEPS)
    assert nfe == nfe2 == 60 and torch.isfinite(sep).all()
    out0 = ops.scale_output(mix_d[:1], sep[:1].contiguous()).cpu()
    d = _rms(out0 - ref)
    r = d / _rms(ref)
    print(f"\n[nf=128 hybrid B=2, row 0 vs the CPU oracle, injected noise] diff rms {d:.3e}  rel {r:.3e}")
    assert d < 1e-3 and r < 1e-2, f"hybrid at nf = 128: {d:.3e} abs / {r:.3e} rel RMS from the oracle"
    e16.close()
    esp.close()


def test_throughput_mode_changes_only_the_grouping_of_partial_sums():
    # evaluate / separate --streams K > 1 and bench.py set engine option rw_quarter (DiffSepModel.set_throughput_mode): a
    # register-weight launch whose blocks would get <= 4 tiles runs on a quarter of the blocks with four times the tiles each.  A
    # block adds the GroupNorm statistics of ITS tiles in fp32 before the (order-independent) integer atomics, so the statistics move
    # in their last bits with the tile range of a block — exactly what a different batch size does (tests/test_round5_gpu.py:
    # 6.4e-3 after 4 evaluations between B = 16 and B = 1) — and a 16-bit trajectory amplifies any last-bit change to its own
    # rounding level: the two results are two realisations of the mode's rounding noise (measured 5.1e-3 relative RMS after these 4
    # evaluations at t = 1 and 0.03, where one realisation is ~3.5e-3 from the fp32 result; B = 16 against B = 1 measures 6.4e-3).  Gate:
    # the batch-independence gate of tests/test_round5_gpu.py (1e-2) — and exactly nothing for the engines that have no register-weight
    # kernel.
    B, T = 16, 32000
    eng, _ = _engine(64, _lib.F16)
    mix = torch.from_numpy(synth.synth_batch(B, T=T)[0]).to(DEV)
    mn, _, _ = ops.normalize_batch(mix)
    kw = dict(N=2, corrector_steps=1, snr=0.5, eps=0.03, denoise=True, seed=5)
    a, _ = eng.pc_sample(mn, SDE, **kw)
    a2, _ = eng.pc_sample(mn, SDE, **kw)
    eng.set_option("rw_quarter", 1)
    assert eng.get_option("rw_quarter") == 1
    b, _ = eng.pc_sample(mn, SDE, **kw)
    b2, _ = eng.pc_sample(mn, SDE, **kw)
    eng.profile_begin()
    eng.pc_sample(mn, SDE, **kw)
    eng.profile_end()
    assert any(r_["kernel"].startswith("conv3x3_rw_kernel") for r_ in eng.profile_records())
    assert torch.equal(a, a2) and torch.equal(b, b2)  # (each mode is deterministic)
    rel = _rms(a - b) / _rms(a)
    print(f"\n[throughput mode vs default, f16 nf=64 B=16, 4 evaluations] rel rms {rel:.3e}")
    assert rel < 1e-2
    eng.close()
    esp, _ = _engine(64, _lib.F32_SPLIT)
    mn4 = mn[:4].contiguous()
    c, _ = esp.pc_sample(mn4, SDE, **kw)
    esp.set_option("rw_quarter", 1)
    d, _ = esp.pc_sample(mn4, SDE, **kw)
    assert torch.equal(c, d)
    esp.close()
    # the model-level switch reaches the engine, also one that is built later
    from diffsep_amd.pl_model import DiffSepModel, default_config
    m = DiffSepModel(default_config(nf=16), dtype="f16").set_throughput_mode(True)
    assert m.score_model.engine().get_option("rw_quarter") == 1
    r = m.replica()
    r.set_throughput_mode(True)
    assert r.score_model.engine() is not m.score_model.engine() and r.score_model.engine().get_option("rw_quarter") == 1
    m.score_model.load_state_dict({k: v * 0.5 for k, v in m.score_model.state_dict().items() if k.startswith("backbone.")})
    assert m.score_model.engine().get_option("rw_quarter") == 1  # (rebuilt from the new weights: the option is re-applied)


def _reference_style_experiment(root, nf=16):
    """exp/<name>/<run>/{hparams.yaml, checkpoints/epoch-xxx.ckpt} + a WSJ0-mix tree, as the reference's training leaves them"""
    import yaml
    from diffsep_amd.pl_model import default_config
    run = root / "exp" / "default" / "2023-01-01_00-00-00_"
    (run / "checkpoints").mkdir(parents=True)
    data = root / "data" / "wsj0_mix"
    for split, n in (("tt", 3), ("cv", 2)):
        base = data / "2speakers" / "wav8k" / "max" / split
        for d_ in ("mix", "s1", "s2"):
            (base / d_).mkdir(parents=True)
        for i in range(n):
            mix, tgt = synth.synth_mixture(10 * (split == "cv") + i, T=26000 + 3000 * i, fs=8000, n_src=2)
            wavio.save(base / "mix" / f"u{i}.wav", torch.from_numpy(mix), 8000)
            for k in range(2):
                wavio.save(base / f"s{k + 1}" / f"u{i}.wav", torch.from_numpy(tgt[k:k + 1]), 8000)
    cfg = default_config(nf=nf)
    ds = lambda split: {"_target_": "datasets.WSJ0_mix", "path": str(data), "n_spkr": 2, "fs": 8000, "cut": "max",
                        "split": split, "max_len_s": None, "max_n_samples": None}
    cfg["datamodule"] = {"train": {"dataset": dict(ds("train"), max_len_s=5)}, "val": {"dataset": ds("val")},
                         "test": {"dataset": ds("test")}}
    with open(run / "hparams.yaml", "w") as f:
        yaml.safe_dump({"config": cfg}, f)
    mcfg = _lib.model_config(nf=nf, num_sources=2)
    table = [(n, s) for n, s, _ in param_table(mcfg)]
    raw, ema = synth.synth_state_dict(table, 1), synth.synth_state_dict(table, 7)
    sd = {"score_model.backbone." + k: torch.from_numpy(v) for k, v in raw.items()}
    sd["score_model.stft.window"] = torch.hann_window(510)
    sd["score_model.stft_inv.window"] = torch.hann_window(510)
    shadow = [torch.from_numpy(ema[n]) for n, _ in table if not n.endswith("all_modules.0.W")]
    ckpt = run / "checkpoints" / "epoch-979_si_sdr-11.111.ckpt"
    torch.save({"state_dict": sd, "hyper_parameters": {"config": cfg}, "ema": {"shadow_params": shadow}}, ckpt)
    return ckpt, run


def test_reference_evaluate_command_line_runs_verbatim(tmp_path, capsys):
    from diffsep_amd import evaluate as ev
    ckpt, run = _reference_style_experiment(tmp_path)
    res = tmp_path / "results"
    # evaluate.py ckpt --test --val -s log -d 0 --save-n 1 -o results   (+ -N 2: a short run; every flag is the reference's)
    out = ev.main([str(ckpt), "--test", "--val", "-s", "log", "-d", "0", "--save-n", "1", "-o", str(res), "-N", "2",
                   "--pesq-mode", "nb", "-w", "2", "-l", "3"])
    # evaluate.py:306-323: <output_dir>/<exp_name>_<ckpt_name>_<tag_inf>
    want = res / f"{run.name}_{ckpt.stem}_N-2_snr-0.5_corrstep-1_denoise-True_schedule-log"
    assert out == want and want.is_dir()
    for split, n in (("test", 3), ("val", 2)):
        rec = json.load(open(want / f"{split}.json"))
        assert [r["batch_idx"] for r in rec] == list(range(n))
        for i, r in enumerate(rec):
            assert set(r) >= {"batch_idx", "si_sdr", "si_sir", "si_sar", "pesq", "stoi", "nfe", "runtime", "len_s"}
            assert r["nfe"] == 4 and abs(r["len_s"] - (26000 + 3000 * i) / 8000) < 1e-9
            assert np.asarray(r["si_sdr"]).shape == (1, 2) and r["pesq"] is None
            assert len(r["stoi"]) == 2 and all(-1.0 <= v <= 1.0 for v in r["stoi"])  # ESTOI per source (evaluate.py:113-130)
        summ = json.load(open(want / f"{split}_summary.json"))
        assert summ["number"] == n and summ["not_computed"] == ["pesq"] and summ["stoi_extended"] is True
        assert abs(summ["stoi"] - np.mean([np.mean(r["stoi"]) for r in rec])) < 1e-12
        # --save-n 1: the first utterance's five files (evaluate.py:70-101), nothing for the others
        wavs = sorted(p.name for p in (want / "wav" / split).glob("*.wav"))
        assert wavs == ["000_enh0.wav", "000_enh1.wav", "000_mix.wav", "000_tgt0.wav", "000_tgt1.wav"]
        peak = max(float(wavio.load(want / "wav" / split / w)[0].abs().max()) for w in wavs)
        assert abs(peak - 0.95) < 1e-6
    # the saved estimates are in the targets' order: enh0 correlates with tgt0 at least as well as with tgt1 ... for the pair
    # the permutation search picked (random-init weights separate nothing: only the bookkeeping is checked)
    rec = json.load(open(want / "test.json"))
    assert sorted(rec[0]["perm"]) == [0, 1]
    # the tag variant (evaluate.py:321-323) and the action check (evaluate.py:230-231)
    out2 = ev.main([str(ckpt), "--test", "--tag", "mytag", "-o", str(res), "-N", "1", "--save-n", "0", "--stoi-no-extended",
                    "-l", "1"])
    assert out2 == res / "mytag_N-1_snr-0.5_corrstep-1_denoise-True_schedule-None"
    assert json.load(open(out2 / "test_summary.json"))["stoi_extended"] is False and not (out2 / "wav").exists()
    with pytest.raises(SystemExit):
        ev.main([str(ckpt), "-o", str(res)])
    assert "No action requested, add --val or --test" in capsys.readouterr().err
    # the EMA weights were the ones evaluated (quirk Q4): the same utterance through DiffSepModel.load_from_checkpoint
    from diffsep_amd.pl_model import DiffSepModel
    m = DiffSepModel.load_from_checkpoint(ckpt)
    shape = dict((n, s_) for n, s_, _ in param_table(m.score_model.cfg))["all_modules.3.weight"]
    assert torch.equal(m.score_model.state_dict()["backbone.all_modules.3.weight"],
                       torch.from_numpy(synth.synth_param("all_modules.3.weight", shape, 7)))


def test_evaluate_no_proc_and_enhance_through_hparams(tmp_path):
    # evaluate.py:245-262: `__no_proc__` scores the unprocessed mixture (nfe 0, runtime 0, folder <output_dir>/mix);
    # evaluate.py:268-271: --enhance reads datamodule.test.dataset of hparams.yaml as a NoisyDataset and keeps the first source
    import yaml
    from diffsep_amd import evaluate as ev
    from diffsep_amd.pl_model import enhancement_config
    ckpt, run = _reference_style_experiment(tmp_path)
    data = tmp_path / "data" / "wsj0_mix"
    out = ev.main(["__no_proc__", "--test", "--dataset-dir", str(data), "--cut", "max", "-o", str(tmp_path / "res"), "--save-n", "0"])
    assert out == tmp_path / "res" / "mix"
    rec = json.load(open(out / "test.json"))
    assert len(rec) == 3 and all(r["nfe"] == 0 and r["runtime"] == 0.0 and np.isfinite(r["si_sdr"]).all() for r in rec)
    # the mixture as the estimate of both sources: SI-SDR of source k = its energy against the other's, they sum to ~0 dB
    assert all(abs(sum(r["si_sdr"][0])) < 3.0 and max(r["si_sdr"][0]) < 10.0 for r in rec)
    assert all(len(r["stoi"]) == 2 and all(0.0 < v < 1.0 for v in r["stoi"]) for r in rec)

    # ---- enhancement experiment: VoiceBank-DEMAND layout, PriorMixSDE model (config/model/nr.yaml), 16 kHz
    vb = tmp_path / "data" / "vctk"
    for d_ in ("noisy", "clean"):
        (vb / "test" / d_).mkdir(parents=True)
    for i in range(2):
        mix, tgt = synth.synth_mixture(20 + i, T=20000 + 4000 * i, fs=16000, n_src=2)
        wavio.save(vb / "test" / "noisy" / f"p{i}.wav", torch.from_numpy(mix), 16000)
        wavio.save(vb / "test" / "clean" / f"p{i}.wav", torch.from_numpy(tgt[:1]), 16000)
    run2 = tmp_path / "exp" / "enh" / "2023-02-02_"
    (run2 / "checkpoints").mkdir(parents=True)
    cfg = enhancement_config(nf=16)
    cfg["datamodule"] = {"test": {"dataset": {"_target_": "datasets.NoisyDataset", "audio_path": str(vb), "fs": 16000,
                                              "split": "test", "audio_len": 4, "augmentation": False}}}
    with open(run2 / "hparams.yaml", "w") as f:
        yaml.safe_dump({"config": cfg}, f)
    mcfg = _lib.model_config(nf=16, num_sources=2, spec_factor=0.15)
    sd = {"score_model.backbone." + n: torch.from_numpy(synth.synth_param(n, s, 7)) for n, s, _ in param_table(mcfg)}
    ck2 = run2 / "checkpoints" / "epoch-10.ckpt"
    torch.save({"state_dict": sd, "hyper_parameters": {"config": cfg}}, ck2)  # (no `ema` entry: the raw weights are used)
    out = ev.main([str(ck2), "--test", "--enhance", "-N", "2", "-o", str(tmp_path / "res"), "--save-n", "1"])
    assert out.name == "2023-02-02__epoch-10_N-2_snr-0.5_corrstep-1_denoise-True_schedule-None"
    rec = json.load(open(out / "test.json"))
    assert len(rec) == 2 and all(np.asarray(r["si_sdr"]).shape == (1, 1) and len(r["perm"]) == 2 and len(r["stoi"]) == 1
                                 and r["nfe"] == 4 for r in rec)
    assert sorted(abs(r["len_s"] - t) < 1e-9 for r, t in zip(sorted(rec, key=lambda r: r["len_s"]), (1.25, 1.5))) == [True, True]
    assert len(list((out / "wav" / "test").glob("000_*.wav"))) == 5
