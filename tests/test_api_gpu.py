"""The reference-shaped Python API on the GPU: DiffSepModel / sdes.get_pc_sampler / predictors /
correctors / separate CLI, all routed to the HIP engine."""
import numpy as np
import pytest
import torch

from diffsep_amd import sdes, synth, wavio
from diffsep_amd.engine import param_table
from diffsep_amd.pl_model import DiffSepModel, default_config
from diffsep_amd import separate as sep_cli

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def rel_rms(a, b):
    a = a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = b.detach().double().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / (np.sqrt(np.mean(b ** 2)) + 1e-30))


class Injected:
    """Feed the reference's RNG call sites (torch.randn_like / torch.randn) from a queue of draws."""
    def __init__(self, draws):
        self.draws, self.i = list(draws), 0

    def __enter__(self):
        self.o1, self.o2 = torch.randn_like, torch.randn

        def nxt(*a, **k):
            z = self.draws[self.i]
            self.i += 1
            return z
        torch.randn_like = lambda x, **k: nxt().to(x.device)
        torch.randn = lambda *a, **k: nxt().to(k.get("device", "cpu"))
        from diffsep_amd.sdes import noise  # (the mirror classes draw through sdes/noise.py, user-written ones through torch)
        self.o3 = noise.set_source(lambda shape, like: nxt().to(like.device))
        return self

    def __exit__(self, *a):
        torch.randn_like, torch.randn = self.o1, self.o2
        from diffsep_amd.sdes import noise
        noise.set_source(self.o3)


def model16(dtype="f32"):
    m = DiffSepModel(default_config(nf=16), dtype=dtype)
    table = [(n, s) for n, s, _ in param_table(m.score_model.cfg)]
    sd = synth.synth_state_dict(table, 7)
    m.score_model.load_state_dict({"backbone." + k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to("cuda:0")


def test_stepwise_api_matches_reference_golden(golden):
    # intermediate=True forces the generic loop: corrector.update_fn / predictor.update_fn objects, each one
    # score evaluation + one fused update kernel, noise drawn at the reference's call sites.
    g, meta = golden
    m = model16()
    B, S, T, N = 2, 2, 4000, 3
    mix = torch.from_numpy(synth.synth_batch(B, T=T)[0]).cuda()
    (mix_norm, _), mean, std = m.normalize_batch((mix, None))
    assert rel_rms(mix_norm, g["g10_mix_norm"]) < 1e-6
    draws = [torch.from_numpy(synth.synth_noise(f"g9.z{i}", (B, S, T))).cuda() for i in range(7)]
    with Injected(draws) as inj:
        sampler = m.get_pc_sampler("reverse_diffusion", "ald2", mix_norm, N=N, denoise=True, intermediate=True,
                                   corrector_steps=1, snr=0.5, schedule=None)
        x, nfe, im = sampler()
        assert inj.i == 7
    assert nfe == meta["g9_nfe"] and len(im) == N
    assert rel_rms(x, g["g9_sep"]) < 1e-4


def test_fused_sampler_is_seeded_by_torch_generator():
    m = model16()
    mix = torch.from_numpy(synth.synth_batch(2, T=4000)[0]).cuda()
    (mix_norm, _), *_ = m.normalize_batch((mix, None))
    torch.manual_seed(3)
    a, nfe = m.get_pc_sampler("reverse_diffusion", "ald2", mix_norm, N=2, corrector_steps=1, snr=0.5)()
    torch.manual_seed(3)
    b, _ = m.get_pc_sampler("reverse_diffusion", "ald2", mix_norm, N=2, corrector_steps=1, snr=0.5)()
    c, _ = m.get_pc_sampler("reverse_diffusion", "ald2", mix_norm, N=2, corrector_steps=1, snr=0.5)()
    assert nfe == 4 and torch.equal(a, b) and not torch.equal(a, c)
    # minibatch splits the batch, scheduled sampler takes N+1 points but still steps 1/N
    d, ns = m.get_pc_sampler("reverse_diffusion", "ald2", mix_norm, N=2, minibatch=1, corrector_steps=1, snr=0.5)()
    assert d.shape == a.shape and ns == [4, 4]
    e, _ = m.get_pc_sampler("reverse_diffusion", "ald2", mix_norm, N=2, schedule="linear", corrector_steps=1)()
    assert torch.isfinite(e).all()
    with pytest.raises(NotImplementedError):
        m.get_pc_sampler("reverse_diffusion", "ald2", mix_norm, N=2, schedule="fib")()


def test_separate_cli_on_wav_folder(tmp_path):
    # BASELINE configs[0]-style plumbing: N wav files in, s0/ s1/ out (separate.py:148-162)
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    for i in range(3):
        wavio.save(ind / f"utt{i}.wav", torch.from_numpy(synth.synth_mixture(i, T=4000)[0]), 8000)
    sep_cli.main([str(ind), str(outd), "--synthetic-weights", "16", "-N", "2", "--dtype", "f32", "--batch", "2"])
    for i in range(3):
        for s in ("s0", "s1"):
            y, sr = wavio.load(outd / s / f"utt{i}.wav")
            assert sr == 8000 and y.shape == (1, 4000) and torch.isfinite(y).all()


def test_separate_cli_streams_do_not_change_the_files(tmp_path):
    # --streams K: K files in flight; with --seed the written wavs are identical for any K
    ind = tmp_path / "in"
    ind.mkdir()
    for i in range(5):
        wavio.save(ind / f"utt{i}.wav", torch.from_numpy(synth.synth_mixture(i, T=4000 + 640 * (i % 2))[0]), 8000)
    outs = []
    for k in (1, 3):
        outd = tmp_path / f"out{k}"
        sep_cli.main([str(ind), str(outd), "--synthetic-weights", "16", "-N", "2", "--seed", "4", "--streams", str(k)])
        outs.append([wavio.load(outd / s / f"utt{i}.wav")[0] for i in range(5) for s in ("s0", "s1")])
    assert all(a.shape == b.shape and torch.equal(a, b) for a, b in zip(*outs))


def test_enhancement_model_api(golden):
    # config/model/nr.yaml shaped model: PriorMixSDE behind the same DiffSepModel / registry surface
    from diffsep_amd.pl_model import enhancement_config
    from diffsep_amd.sdes import PriorMixSDE
    g, _ = golden
    cfg = enhancement_config(nf=16)
    cfg["model"]["score_model"]["spec_factor"] = 0.33  # the golden weights were generated with the default factor
    m = DiffSepModel(cfg, dtype="f32")
    assert isinstance(m.sde, PriorMixSDE) and m.sde.avg_len == 510
    table = [(n, s) for n, s, _ in param_table(m.score_model.cfg)]
    sd = synth.synth_state_dict(table, 7)
    m.score_model.load_state_dict({"backbone." + k: torch.from_numpy(v) for k, v in sd.items()})
    B, S, T, N = 2, 2, 4000, 3
    mix_norm = torch.from_numpy(g["g10_mix_norm"]).cuda()
    draws = [torch.from_numpy(synth.synth_noise(f"g9.z{i}", (B, S, T))).cuda() for i in range(7)]
    with Injected(draws):
        x, nfe, im = m.get_pc_sampler("reverse_diffusion", "ald2", mix_norm, N=N, denoise=True, intermediate=True,
                                      corrector_steps=1, snr=0.5, schedule=None)()
    assert nfe == 6 and rel_rms(x, g["g11_sep"]) < 1e-4


def test_evaluate_cli_writes_reference_style_results(tmp_path):
    # evaluate.py counterpart: sampler + synced timing + SI-SDR, json records {batch_idx, si_sdr, nfe, runtime, len_s}
    import json
    from diffsep_amd import evaluate as ev
    ev.main(["--synthetic", "3", "--samples", "4000", "--synthetic-weights", "16", "-N", "2", "--dtype", "f32", "--flat-output", "-o",
             str(tmp_path / "sep")])
    rec = json.load(open(tmp_path / "sep" / "test.json"))
    assert [r["batch_idx"] for r in rec] == [0, 1, 2]
    assert all(r["nfe"] == 4 and r["runtime"] > 0 and abs(r["len_s"] - 0.5) < 1e-9 and np.isfinite(r["si_sdr"]).all()
               and np.asarray(r["si_sdr"]).shape == (1, 2) for r in rec)   # per-source lists (evaluate.py:394-405)
    summ = json.load(open(tmp_path / "sep" / "test_summary.json"))
    assert summ["number"] == 3 and summ["world_size"] == 1 and summ["nfe"] == 4
    assert abs(summ["si_sdr"] - np.mean([np.mean(r["si_sdr"]) for r in rec])) < 1e-9
    # --enhance: PriorMixSDE model (nr.yaml); both channels (speech, noise) are scored with the best permutation and
    # the first n_src = 1 entry is kept (evaluate.py:105-127,268-271)
    ev.main(["--synthetic", "2", "--samples", "4000", "--synthetic-weights", "16", "-N", "2", "--dtype", "f32",
             "--enhance", "--flat-output", "-o", str(tmp_path / "enh")])
    rec = json.load(open(tmp_path / "enh" / "test.json"))
    assert len(rec) == 2 and all(np.asarray(r["si_sdr"]).shape == (1, 1) and len(r["perm"]) == 2 for r in rec)
    assert all(abs(r["si_sir"][0][0]) < 99.0 for r in rec)  # a real interference term (one channel alone would clamp)
    assert all({"batch_idx", "si_sdr", "si_sir", "si_sar", "pesq", "stoi", "nfe", "runtime", "len_s"} <= set(r) for r in rec)


def test_evaluate_streams_do_not_change_results(tmp_path):
    # --streams K keeps K utterances in flight; the records must not depend on K (seeds are drawn per utterance)
    import json
    from diffsep_amd import evaluate as ev
    recs = []
    for k in (1, 3):
        ev.main(["--synthetic", "7", "--samples", "6000", "--synthetic-weights", "16", "-N", "2", "--streams", str(k),
                 "--flat-output", "-o", str(tmp_path / f"k{k}")])
        recs.append(json.load(open(tmp_path / f"k{k}" / "test.json")))
        summ = json.load(open(tmp_path / f"k{k}" / "test_summary.json"))
        assert summ["streams"] == k and summ["number"] == 7 and summ["utt_per_s_rank0"] > 0
    strip = lambda r: {k: v for k, v in r.items() if k != "runtime"}
    assert [strip(r) for r in recs[0]] == [strip(r) for r in recs[1]]


def test_evaluate_streams_on_files_of_different_length(tmp_path):
    # a flat mix/ s1/ s2/ folder with utterances of different lengths: every utterance is a new workspace plan and a new
    # graph capture on its worker's stream; the records must not depend on the number of streams
    import json
    from diffsep_amd import evaluate as ev
    root = tmp_path / "data"
    for sub in ("mix", "s1", "s2"):
        (root / sub).mkdir(parents=True)
    for i in range(7):
        mix, tgt = synth.synth_mixture(i, T=3000 + 517 * ((i * 3) % 7), fs=8000, n_src=2)
        wavio.save(root / "mix" / f"u{i}.wav", torch.from_numpy(mix), 8000)
        for k in range(2):
            wavio.save(root / f"s{k + 1}" / f"u{i}.wav", torch.from_numpy(tgt[k:k + 1]), 8000)
    assert wavio.info(root / "mix" / "u1.wav") == (8000, 3000 + 517 * 3)
    recs = []
    for k in (1, 3):
        ev.main(["--dataset-dir", str(root), "--synthetic-weights", "16", "-N", "2", "--streams", str(k), "--flat-output", "-o",
                 str(tmp_path / f"k{k}")])
        recs.append(json.load(open(tmp_path / f"k{k}" / "test.json")))
    strip = lambda r: {k: v for k, v in r.items() if k != "runtime"}
    assert len(recs[0]) == 7 and [strip(r) for r in recs[0]] == [strip(r) for r in recs[1]]
    assert [r["len_s"] for r in recs[0]] == [(3000 + 517 * ((i * 3) % 7)) / 8000 for i in range(7)]


def test_evaluate_cli_on_wsj0_mix_tree(tmp_path):
    # WSJ0-mix on-disk layout -> evaluate: file order, (mix, tgt) shapes, variable lengths
    import json
    from diffsep_amd import evaluate as ev, wavio
    base = tmp_path / "wsj" / "2speakers" / "wav8k" / "min" / "tt"
    for d in ("mix", "s1", "s2"):
        (base / d).mkdir(parents=True)
    for i, T in enumerate((4000, 3000)):
        mix, tgt = synth.synth_mixture(i, T=T, fs=8000, n_src=2)
        wavio.save(base / "mix" / f"u{i}.wav", torch.from_numpy(mix) * 0.5, 8000)
        for k in range(2):
            wavio.save(base / f"s{k + 1}" / f"u{i}.wav", torch.from_numpy(tgt[k:k + 1]) * 0.5, 8000)
    ev.main(["--dataset-dir", str(tmp_path / "wsj"), "--cut", "min", "--split", "test", "--synthetic-weights", "16",
             "-N", "2", "--dtype", "f32", "--flat-output", "-o", str(tmp_path / "out")])
    rec = json.load(open(tmp_path / "out" / "test.json"))
    assert [abs(r["len_s"] - t) < 1e-9 for r, t in zip(rec, (0.5, 0.375))] == [True, True]
    assert json.load(open(tmp_path / "out" / "test_summary.json"))["number"] == 2
