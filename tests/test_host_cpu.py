"""Host-side logic that needs no GPU: registries, config access, wav I/O, checkpoint -> weight mapping with
the EMA swap, utterance sharding and the multi-rank result gather (gloo, world_size 2)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffsep_amd import dist_utils, synth, wavio
from diffsep_amd.engine import param_table
from diffsep_amd.pl_model import DiffSepModel, cfg_get, default_config
from diffsep_amd.sdes import CorrectorRegistry, PredictorRegistry, SDERegistry, MixSDE


def test_registries_expose_reference_names():
    assert {"reverse_diffusion", "euler_maruyama", "none"} == set(PredictorRegistry.get_all_names())
    assert {"ald2", "ald", "langevin", "none"} == set(CorrectorRegistry.get_all_names())  # = the reference's set
    assert set(SDERegistry.get_all_names()) >= {"mix", "priormix"}
    assert SDERegistry.get_by_name("mix") is MixSDE
    with pytest.raises(ValueError):
        PredictorRegistry.get_by_name("nope")


def test_cfg_get_on_dicts_and_objects():
    cfg = default_config(nf=16)
    assert cfg_get(cfg, "model.sampler.N") == 30 and cfg_get(cfg, "model.missing.x", 7) == 7

    class O:
        pass
    o = O(); o.model = O(); o.model.fs = 16000
    assert cfg_get(o, "model.fs") == 16000


def test_mixsde_copy_and_eigenvalues():
    s = MixSDE(2, 2.0, 0.05, 0.5, N=30)
    c = s.copy(); c.N = 7
    assert s.N == 30 and c.N == 7 and c.T == 1.0
    ev1, ev2 = s._cov_eigval(torch.tensor([1.0]))
    assert abs(float(ev1) - 0.2475) < 1e-6 and abs(float(ev2) - 0.1337663) < 1e-6


def test_wav_roundtrip(tmp_path):
    x = torch.from_numpy(synth.synth_mixture(0, T=4000)[0])
    wavio.save(tmp_path / "a.wav", x, 8000)
    y, sr = wavio.load(tmp_path / "a.wav")
    assert sr == 8000 and y.shape == x.shape and float((x - y).abs().max()) < 1.0 / 16384
    wavio.save(tmp_path / "b.wav", x, 8000, bits=32)
    z, _ = wavio.load(tmp_path / "b.wav")
    assert torch.equal(z, x)
    with pytest.raises(ValueError):
        (tmp_path / "c.wav").write_bytes(b"not a wav file at all")
        wavio.load(tmp_path / "c.wav")


def test_checkpoint_loader_applies_ema(tmp_path):
    cfg = default_config(nf=16)
    m = DiffSepModel(cfg)
    names = [n for n, _, _ in param_table(m.score_model.cfg)]
    raw = synth.synth_state_dict([(n, s) for n, s, _ in param_table(m.score_model.cfg)], 1)
    ema = synth.synth_state_dict([(n, s) for n, s, _ in param_table(m.score_model.cfg)], 2)
    sd = {"score_model.backbone." + n: torch.from_numpy(v) for n, v in raw.items()}
    sd["score_model.stft.window"] = torch.hann_window(510)
    shadow = [torch.from_numpy(ema[n]) for n in names if not n.endswith("all_modules.0.W")]  # torch_ema skips frozen W
    torch.save({"state_dict": sd, "hyper_parameters": {"config": cfg}, "ema": {"shadow_params": shadow}},
               tmp_path / "m.ckpt")
    m2 = DiffSepModel.load_from_checkpoint(tmp_path / "m.ckpt")
    st = {k[len("backbone."):]: v.numpy() for k, v in m2.score_model.state_dict().items() if k.startswith("backbone.")}
    assert np.array_equal(st["all_modules.0.W"], raw["all_modules.0.W"])  # frozen Fourier weights: no EMA
    assert np.array_equal(st["all_modules.3.weight"], ema["all_modules.3.weight"])
    assert np.array_equal(st["output_layer.bias"], ema["output_layer.bias"])
    # what an engine would be created from = the packed EMA weights
    from diffsep_amd.engine import pack_state_dict
    want = dict(ema)
    want["all_modules.0.W"] = raw["all_modules.0.W"]
    assert np.array_equal(m2.score_model.packed_blob(), pack_state_dict(m2.score_model.cfg, want))
    m3 = DiffSepModel.load_from_checkpoint(tmp_path / "m.ckpt", use_ema=False)
    assert np.array_equal(m3.score_model.state_dict()["backbone.all_modules.3.weight"].numpy(), raw["all_modules.3.weight"])
    with pytest.raises(RuntimeError):  # (strict, like torch: the reference's load_from_checkpoint is)
        m.score_model.load_state_dict({})


def test_shard_range_matches_reference_partition():
    # evaluate_mp.py:495-503: floor(n/workers) per worker, last one takes the remainder
    for n, w in ((3000, 8), (10, 3), (7, 7), (5, 8), (0, 2)):
        rs = [dist_utils.shard_range(n, w, r) for r in range(w)]
        assert rs[0][0] == 0 and rs[-1][1] == n
        assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
        assert all(e - s == n // w for s, e in rs[:-1])
    parts = dist_utils.length_balanced_order([5, 1, 9, 3, 7, 2], 2)
    assert sorted(parts[0] + parts[1]) == list(range(6))
    # evaluate --balance: every utterance exactly once, per-rank work within one utterance of each other
    lens = [3 + (i * 37) % 11 for i in range(50)]
    for w in (1, 3, 8):
        got = [dist_utils.rank_indices(50, w, r, lens, balance=True) for r in range(w)]
        assert sorted(i for g in got for i in g) == list(range(50)) and all(g == sorted(g) for g in got)
        loads = [sum(lens[i] for i in g) for g in got]
        assert max(loads) - min(loads) <= max(lens)
        assert [dist_utils.rank_indices(50, w, r, lens) for r in range(w)] == \
               [list(range(*dist_utils.shard_range(50, w, r))) for r in range(w)]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = dist_utils.shard_range(5, world, rank)
    local = [torch.full((2, 100 + 10 * i), float(i)) for i in range(lo, hi)]  # variable-length "waveforms"
    out = dist_utils.gather_waveforms(local, None, device=torch.device("cpu"))
    objs = dist_utils.gather_objects([{"batch_idx": i} for i in range(lo, hi)])
    if rank == 0:
        q.put(([(tuple(o.shape), float(o[0, 0])) for o in out], objs))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_across_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, objs = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == [((2, 100 + 10 * i), float(i)) for i in range(5)]
    assert [r["batch_idx"] for part in objs for r in part] == list(range(5))


def test_wsj0_mix_and_noisy_dataset_layouts(tmp_path):
    # datasets/wsj0_mix.py:64-92 and datasets/vctk_demand.py:33-61 on-disk contracts
    from diffsep_amd import datasets, wavio
    base = tmp_path / "3speakers" / "wav16k" / "max" / "cv"
    for d in ("mix", "s1", "s2", "s3"):
        (base / d).mkdir(parents=True)
    g = torch.Generator().manual_seed(0)
    lens = {"b.wav": 900, "a.wav": 1200}
    for name, T in lens.items():
        src = (torch.rand(3, T, generator=g) - 0.5) * 0.4
        for k in range(3):
            wavio.save(base / f"s{k + 1}" / name, src[k:k + 1], 16000)
        wavio.save(base / "mix" / name, src.sum(0, keepdim=True), 16000)
    ds = datasets.WSJ0_mix(tmp_path, n_spkr=3, fs=16000, cut="max", split="val")
    assert ds.file_list == ["a.wav", "b.wav"] and len(ds) == 2
    mix, tgt = ds[0]
    assert mix.shape == (1, 1200) and tgt.shape == (3, 1200)
    assert (mix - tgt.sum(0, keepdim=True)).abs().max() < 3e-4  # 16-bit quantisation
    assert len(datasets.WSJ0_mix(tmp_path, n_spkr=3, fs=16000, split="val", max_n_samples=1)) == 1
    assert ds.num_samples(0) == 1200 and ds.num_samples(1) == 900  # from the wav headers alone
    with pytest.raises(NotImplementedError):  # random training crops are not part of the inference path
        datasets.WSJ0_mix(tmp_path, n_spkr=3, fs=16000, split="val", max_len_s=0.05)
    with pytest.raises(ValueError):  # a corpus opened at the wrong rate must not be read silently
        datasets.WSJ0_mix(tmp_path, n_spkr=3, fs=16000, split="val").__class__(
            ds.mix_dir, ds.target_dirs, ds.file_list, 8000)[0]
    for bad in (dict(fs=44100), dict(n_spkr=4), dict(cut="mid"), dict(split="dev")):
        with pytest.raises(ValueError):
            datasets.WSJ0_mix(tmp_path, **{**dict(n_spkr=3, fs=16000, split="val"), **bad})
    mb, tb = datasets.max_collator([ds[0], ds[1]])
    assert mb.shape == (2, 1, 1200) and tb.shape == (2, 3, 1200)
    assert mb[1, 0, :150].abs().max() == 0 and mb[1, 0, 150] == ds[1][0][0, 0]  # centre padding
    mr, tr_, lens_ = datasets.pad_batch([ds[0], ds[1]], side="right")               # the engine's mixed-length batches
    assert lens_ == [1200, 900] and mr[1, 0, 0] == ds[1][0][0, 0] and mr[1, 0, 900:].abs().max() == 0
    for split in ("train", "test"):
        for d in ("noisy", "clean"):
            (tmp_path / "vb" / split / d).mkdir(parents=True)
        clean = (torch.rand(1, 700, generator=g) - 0.5) * 0.4
        noisy = clean + (torch.rand(1, 700, generator=g) - 0.5) * 0.1
        wavio.save(tmp_path / "vb" / split / "clean" / "p1.wav", clean, 16000)
        wavio.save(tmp_path / "vb" / split / "noisy" / "p1.wav", noisy, 16000)
    noisy_t, tgt_t = datasets.NoisyDataset(tmp_path / "vb", split="test")[0]
    assert noisy_t.shape == (1, 700) and tgt_t.shape == (2, 700)
    assert torch.equal(tgt_t[0:1] + tgt_t[1:2], noisy_t) or (tgt_t.sum(0, keepdim=True) - noisy_t).abs().max() < 1e-6
    noisy_tr, tgt_tr = datasets.voicebank_demand(tmp_path / "vb", split="train")[0]  # whole utterances through the resolver
    assert noisy_tr.shape == (1, 700) and tgt_tr.shape == (2, 700)
    with pytest.raises(NotImplementedError):  # the reference's default split serves random training crops
        datasets.NoisyDataset(tmp_path / "vb")
    with pytest.raises(NotImplementedError):
        datasets.NoisyDataset(tmp_path / "vb", augmentation=True, split="test")
    with pytest.raises(ValueError):
        datasets.NoisyDataset(tmp_path / "vb", split="val")


def test_summarize_matches_reference_schema():
    from diffsep_amd.datasets import summarize
    s = summarize([{"si_sdr": 1.0, "pesq": None, "nfe": 60, "x": [1.0, 3.0]}, {"si_sdr": 3.0, "pesq": None, "nfe": 60, "x": [3.0, 5.0]}])
    assert s == {"si_sdr": 2.0, "nfe": 60.0, "x": 3.0, "number": 2}


def test_bench_multi_rank_sequencing_gloo(tmp_path):
    # bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per GPU), here with 2 CPU ranks
    # over gloo and a stand-in engine (DIFFSEP_BENCH_DRYRUN=1): every rank must enter the same collectives (the
    # untimed roofline pass of rank 0 must not), rank 0 prints exactly one JSON line, nobody hangs.
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, DIFFSEP_BENCH_DRYRUN="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
                                       "--warmup", "1", "--batch", "2", "--samples", "800"], env=e, cwd=root,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-500:] for o in outs]
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["scaling"] == "weak" and "roofline" in res
    assert res["config"]["sharding"] == "utterances/2" and res["value"] > 0
    assert res["ranks_seen"] == [[0, 0], [1, 1]]
    # weak mode carries every rank's own time, their imbalance and the per-rank engine / memory situation too
    assert len(res["rank_elapsed_s_per_step"]) == 2 and res["imbalance_max_over_mean"] >= 1.0
    assert res["hbm"]["engines_per_rank"] == [4, 4]


def test_bench_self_launches_its_ranks_gloo():
    # `python bench.py --gpus 2` with NO torch.distributed.run around it (how the round-2 driver invoked it): bench.py
    # must start the two ranks itself, and the single JSON line must say n_gpus = 2 with both ranks seen
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["DIFFSEP_BENCH_DRYRUN"] = "1"
    for extra in ([], ["--scaling", "strong", "--utterances", "7"], ["--in-flight", "1"]):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                              "--batch", "2", "--samples", "800"] + extra, env=env, cwd=root, capture_output=True,
                             text=True, timeout=240)
        assert out.returncode == 0, out.stderr[-800:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, out.stdout[-500:]
        res = json.loads(lines[0])
        assert res["n_gpus"] == 2 and res["value"] > 0 and res["ranks_seen"] == [[0, 0], [1, 1]]
        assert res["scaling"] == ("strong" if "strong" in extra else "weak") and res["imbalance_max_over_mean"] >= 1.0
        if "--in-flight" in extra:
            assert res["config"]["batches_in_flight"] == 1 and res["hbm"]["engines_per_rank"] == [1, 1]


def test_bench_strong_scaling_sequencing_gloo():
    # bench.py --scaling strong as the driver would launch it on 2 GPUs, with 2 CPU ranks over gloo and the stand-in
    # engine: a fixed set split by length, width-bucketed batches, ONE gather per pass, one JSON line with the per-rank
    # busy times and their imbalance
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, DIFFSEP_BENCH_DRYRUN="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
                                       "--warmup", "1", "--batch", "4", "--scaling", "strong", "--utterances", "11"],
                                      env=e, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-800:] for o in outs]
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")]
    res = json.loads(lines[0])
    assert res["scaling"] == "strong" and res["n_gpus"] == 2 and res["config"]["utterances"] == 11
    assert len(res["rank_busy_s_per_step"]) == 2 and res["imbalance_max_over_mean"] >= 1.0 and res["value"] > 0
    assert res["ranks_seen"] == [[0, 0], [1, 1]]


def test_evaluate_cli_two_ranks_balanced_gloo(tmp_path):
    # `python -m diffsep_amd.evaluate --balance` end to end on 2 CPU ranks over gloo, the device replaced by the stand-ins
    # of tests/evaluate_gloo_rank.py: both ranks separate a share, the gathered <split>.json holds every utterance
    # once, in order, and each record equals the one a single rank computes (seeds and batching do not depend on the
    # world size or on the deal); the balanced deal differs from the contiguous one
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    argv = ["--synthetic", "13", "--samples", "6000", "--samples-max", "30000", "--synthetic-weights", "16", "-N", "3",
            "--batch", "3", "--streams", "2", "--seed", "5"]

    def run(world, extra, out):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs = []
        for r in range(world):
            e = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world)) if world > 1 else env
            procs.append(subprocess.Popen([sys.executable, os.path.join(here, "evaluate_gloo_rank.py")] + argv + extra +
                                          ["--flat-output", "-o", str(out)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        outs = [p.communicate(timeout=240) for p in procs]
        assert all(p.returncode == 0 for p in procs), [o[1][-800:] for o in outs]
        calls = [eval([l for l in o[0].splitlines() if l.startswith("CALLS")][0].split(" ", 2)[2]) for o in outs]
        return json.load(open(out / "test.json")), json.load(open(out / "test_summary.json")), calls

    one, s1, c1 = run(1, [], tmp_path / "w1")
    # half precision can overflow where fp32 cannot: a batch with non-finite samples is repeated on the split-precision twin
    os.environ["EVAL_TEST_OVERFLOW"] = "1"
    try:
        ovf, s0, _ = run(1, [], tmp_path / "w1o")
    finally:
        del os.environ["EVAL_TEST_OVERFLOW"]
    assert s0["split_fallback_batches_rank0"] >= 1 and s1["split_fallback_batches_rank0"] == 0
    assert [{k: v for k, v in r.items() if k != "runtime"} for r in ovf] == [{k: v for k, v in r.items() if k != "runtime"} for r in one]
    bal, s2, c2 = run(2, ["--balance"], tmp_path / "w2b")
    con, s3, c3 = run(2, [], tmp_path / "w2c")
    strip = lambda rs: [{k: v for k, v in r.items() if k != "runtime"} for r in rs]
    assert [r["batch_idx"] for r in bal] == list(range(13)) == [r["batch_idx"] for r in con]
    assert strip(bal) == strip(one) == strip(con)
    assert s2["world_size"] == 2 and s2["number"] == 13 and s1["world_size"] == 1 and s2["si_sdr"] == s1["si_sdr"]
    # every rank separated something in both deals; the timed calls (warm-up = K calls of the first batch excluded)
    for calls in (c2, c3):
        assert all(len(c) > 2 for c in calls)
        assert sum(b for c in calls for b, _ in c[2:]) == 13
    # contiguous ranges give rank 1 the remainder (7 of 13), the balanced deal alternates by length (7 / 6)
    assert [sum(b for b, _ in c[2:]) for c in c3] == [6, 7] and [sum(b for b, _ in c[2:]) for c in c2] == [7, 6]
    assert c2 != c3


def test_model_dtype_auto_picks_by_width():
    # dtype="auto" (the default of DiffSepModel and of both CLIs): f16 up to nf = 64, hybrid (split head + f16) for wider
    # backbones, whose 16-bit rounding costs more agreement with fp32 (tests/test_fullsize_gpu.py)
    from diffsep_amd.pl_model import DiffSepModel, default_config, enhancement_config, HYBRID_HEAD_STEPS
    m = DiffSepModel(default_config(nf=16))
    assert m.dtype == "f16" and m.tail_model is None
    m = DiffSepModel(default_config(nf=64), dtype="auto")
    assert m.dtype == "f16"
    m = DiffSepModel(enhancement_config(nf=128))
    assert m.dtype == "hybrid" and m.tail_model is not None and m.head_steps == HYBRID_HEAD_STEPS
    assert m.tail_model.lib_kind == "f16" and m.score_model.cfg.dtype != m.tail_model.cfg.dtype
    assert DiffSepModel(default_config(nf=128), dtype="f16").dtype == "f16"


def test_plan_batches_buckets_by_padded_width():
    from diffsep_amd.evaluate import plan_batches
    width = lambda T: 64 * ((1 + (T + 382) // 128 + 63) // 64)
    lengths = [32000, 31000, 40000, 8000, 32001, 7000, 39000, 31999]
    got = plan_batches(range(len(lengths)), lengths, width, 3)
    assert got == [[5], [3], [4, 0, 7], [1], [2, 6]]          # ascending width, longest first, <= 3 per call
    assert all(len({width(lengths[i]) for i in g}) == 1 for g in got)
    assert sorted(i for g in got for i in g) == list(range(len(lengths)))


def test_overflow_fallback_follows_weights_and_device():
    # the split fallback of an f16 model must never run on the weights / device of the moment it was made.
    # (i) whether a fallback exists is decided from the mode, nothing is constructed by get_pc_sampler's wrapper;
    # (ii) a twin re-reads its parent's weights and device when the parent changed after the twin was made.
    m = DiffSepModel(default_config(nf=16), dtype="f16")
    assert m.has_fallback() and not DiffSepModel(default_config(nf=16), dtype="bf16").has_fallback()
    assert not DiffSepModel(default_config(nf=16), dtype="split").has_fallback()
    assert getattr(m, "_fallback", None) is None
    fb = m.fallback_model()
    tw = fb.score_model
    assert tw.owner is m.score_model and tw.cfg.dtype != m.score_model.cfg.dtype

    # engines are device objects: a recording stand-in shows what each one WOULD be created from
    built = []

    class FakeEngine:
        def __init__(self, cfg, blob, device=None, lib_kind=None):
            self.blob, self.device, self.dtype, self.lib_kind = blob.copy(), device, cfg.dtype, lib_kind
            built.append(self)

        def close(self):
            pass

    from diffsep_amd.engine import pack_state_dict
    m.score_model._engine_factory = FakeEngine
    e0, t0 = m.score_model.engine(), tw.engine()
    assert e0 is m.score_model.engine() and t0 is tw.engine() and len(built) == 2   # (unchanged weights: nothing rebuilt)
    np.testing.assert_array_equal(e0.blob, t0.blob)
    # new weights on the parent (directly on the score model, as a user of the reference API would): both engines follow
    sd = {k: v + 1.0 for k, v in m.score_model.state_dict().items() if k.startswith("backbone.")}
    m.score_model.load_state_dict(sd)
    t1 = tw.engine()
    assert t1 is not t0 and len(built) == 3
    np.testing.assert_array_equal(t1.blob, pack_state_dict(tw.cfg, {k[len("backbone."):]: v for k, v in sd.items()}))
    assert m.score_model.engine() is not e0 and len(built) == 4
    # the same values loaded again: the content decides, nothing is rebuilt
    m.score_model.load_state_dict(sd)
    assert tw.engine() is t1 and len(built) == 4
    # a write through .data (what torch_ema.copy_to does — invisible to Tensor._version) after eval() / train()
    m.score_model.eval()
    with torch.no_grad():
        m.score_model.backbone.output_layer.bias.data.copy_(torch.full((4,), 0.25))
    t2 = tw.engine()
    assert t2 is not t1 and np.all(t2.blob[24:28] == 0.25)
    # an in-place write on the parameter itself (optimiser step, p.copy_) is seen without any hook
    with torch.no_grad():
        m.score_model.backbone.output_layer.bias.add_(1.0)
    assert np.all(tw.engine().blob[24:28] == 1.25)
    # ... and the device
    m.to("cuda:1")
    assert tw.engine().device == "cuda:1" and m.score_model.engine().device == "cuda:1"
    # DiffSepModel.load_state_dict drops the cached fallback object as well
    m.load_state_dict(sd)
    assert m._fallback is None and m.fallback_model().score_model.owner is m.score_model
    # a hybrid model's head engine and fallback are one twin of its score model: one set of parameters
    h = DiffSepModel(default_config(nf=16), dtype="hybrid")
    h.score_model._engine_factory = FakeEngine
    sdh = {k: v * 0.5 for k, v in h.score_model.state_dict().items() if k.startswith("backbone.")}
    h.load_state_dict(sdh)
    assert h.fallback_model().score_model is h.tail_model and h.tail_model.owner is h.score_model
    np.testing.assert_array_equal(h.tail_model.engine().blob,
                                  pack_state_dict(h.tail_model.cfg, {k[len("backbone."):]: v for k, v in sdh.items()}))
    assert h.tail_model.engine().lib_kind == "f16" and h.tail_model.engine().dtype != h.score_model.engine().dtype


def test_bench_eight_ranks_weak_and_strong_gloo():
    # the 8-rank shape of the driver's scaling run, as a gloo dry run (stand-in engine): every
    # rank takes part (ranks_seen has 8 entries), weak mode keeps 16-per-rank semantics, strong mode deals a fixed set with
    # a remainder (19 utterances on 8 ranks) and gathers every utterance exactly once.
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["DIFFSEP_BENCH_DRYRUN"] = "1"
    env["OMP_NUM_THREADS"] = "1"
    for extra in ([], ["--scaling", "strong", "--utterances", "19"]):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1",
                              "--batch", "2", "--samples", "800", "--in-flight", "1"] + extra, env=env, cwd=root,
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-800:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, out.stdout[-500:]
        res = json.loads(lines[0])
        assert res["n_gpus"] == 8 and res["ranks_seen"] == [[r, r] for r in range(8)] and res["value"] > 0
        if extra:
            assert res["scaling"] == "strong" and res["config"]["utterances"] == 19
            assert len(res["rank_busy_s_per_step"]) == 8
            assert res["gathered_utterances"] == 19 and res["gathered_unique"] == 19
        else:
            assert res["scaling"] == "weak" and res["config"]["sharding"] == "utterances/8"
            assert len(res["rank_elapsed_s_per_step"]) == 8


class _RecordingEngine:  # (module level: picklable)
    def __init__(self, cfg, blob, device=None, lib_kind=None):
        self.blob = blob.copy()

    def close(self):
        pass

    def set_option(self, name, value):
        pass


def test_score_model_module_survives_deepcopy_pickle_and_half():
    # what PyTorch / Lightning utilities do to a module that the reference's LightningModule holds: copy.deepcopy, torch.save of the
    # whole module, .half().  An engine is a device object of this process: copies get an empty slot and build their own engine from
    # their own parameters.
    import copy
    import io
    from diffsep_amd.score_models import ScoreModelNCSNpp
    m = ScoreModelNCSNpp(2, dict(n_fft=510, hop_length=128, center=True, pad_mode="constant"), dict(nf=16))
    m._engine_factory = _RecordingEngine
    b0 = m.engine().blob.copy()
    m2 = copy.deepcopy(m)
    assert m2._slot._engine is None and m2._slot.owner is m2 and m2.twin("split").owner is m2
    np.testing.assert_array_equal(m2.engine().blob, b0)
    with torch.no_grad():
        m2.backbone.output_layer.bias.add_(1.0)
    assert np.all(m2.engine().blob[24:28] == b0[24:28] + 1.0)
    np.testing.assert_array_equal(m.engine().blob, b0)           # (the original is untouched)
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m3 = torch.load(buf, weights_only=False)
    assert m3._slot._engine is None
    np.testing.assert_array_equal(m3.engine().blob, b0)
    m.half()                                                     # parameters AND the window buffers are rounded; the engine follows
    b1 = m.engine().blob
    assert b1.dtype == np.float32 and 0 < float(np.abs(b1 - b0).max()) < 2e-2
    sd = m.state_dict()
    assert sd["stft.window"].dtype == torch.float16 and len(sd) == 647 + 2
