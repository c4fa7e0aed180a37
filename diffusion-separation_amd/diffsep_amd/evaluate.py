"""evaluate.py — the sampler + timing part of the reference's evaluate.py / evaluate_mp.py
(evaluate.py:322-443, evaluate_mp.py:154-326,495-528) on the HIP engine, one rank per GPU.

    python -m diffsep_amd.evaluate --synthetic 32 --synthetic-weights 64 -o out/            (1 GPU)
    python -m torch.distributed.run --nproc-per-node 8 -m diffsep_amd.evaluate ...          (8 GPUs)

Utterances are sharded over ranks in contiguous ranges (evaluate_mp.py:495-503); each rank separates its
share, records {batch_idx, si_sdr, nfe, runtime, len_s} per utterance (evaluate.py:394-405; runtime is
measured WITH a device sync, unlike evaluate.py:374-376) and rank 0 gathers everything (RCCL) and writes
<split>.json + <split>_summary.json (evaluate.py:436-443).  Dataset: --dataset-dir ROOT in the WSJ0-mix layout
(datasets/wsj0_mix.py:64-92; with --enhance the VoiceBank-DEMAND layout, datasets/vctk_demand.py:33-36), a flat
ROOT/{mix,s1,s2} folder, or --synthetic N speech-like mixtures.  SI-SDR (scale-invariant
SDR with the best source permutation) is computed in the normalised domain like evaluate.py:360,382.
"""
import argparse
import json
import os
import time
from pathlib import Path

import torch
import torch.distributed as dist

from . import datasets, metrics, synth, wavio
from .dist_utils import gather_objects, rank_indices
from .pl_model import DiffSepModel, cfg_get, default_config, enhancement_config


def compute_metrics(est, ref):
    """est, ref [S,T] -> dict with the reference's record fields (evaluate.py:103-132): si_sdr / si_sir / si_sar
    (mean over sources at the best permutation), per-source SI-SDR and the permutation.  The waveform reductions run
    in the HIP Gram kernel."""
    sdr, sir, sar, perm = metrics.si_bss_eval_sources(ref[None], est[None])
    return {"si_sdr": float(sdr.mean()), "si_sir": float(sir.mean()), "si_sar": float(sar.mean()),
            "si_sdr_per_source": [float(v) for v in sdr[0]], "perm": [int(v) for v in perm[0]]}


def load_dataset(args, fs):
    if args.dataset_dir and args.enhance:
        ds = datasets.NoisyDataset(args.dataset_dir, fs=fs, split=args.split)
        n = len(ds) if args.limit is None else min(len(ds), args.limit)
        return n, (lambda i: tuple(t[..., : min(ds[i][0].shape[-1], ds[i][1].shape[-1])] for t in ds[i])), None
    if args.dataset_dir:
        root = Path(args.dataset_dir)
        if (root / "mix").is_dir():  # flat folder: mix/, s1/, s2/ ...
            names = sorted(p.name for p in (root / "mix").glob("*.wav"))[: args.limit]

            def get_flat(i):
                mix, _ = wavio.load(root / "mix" / names[i])
                tgt = torch.cat([wavio.load(root / f"s{k + 1}" / names[i])[0][:1] for k in range(args.n_speakers)], 0)
                return mix[:1], tgt
            return len(names), get_flat, [wavio.info(root / "mix" / nm)[1] for nm in names]
        ds = datasets.WSJ0_mix(root, n_spkr=args.n_speakers, fs=fs, cut=args.cut, split=args.split,
                               max_n_samples=args.limit)
        return len(ds), (lambda i: ds[i]), [wavio.info(ds.path_mix / nm)[1] for nm in ds.file_list]
    n = args.synthetic

    def get(i):
        mix, tgt = synth.synth_mixture(i, T=args.samples, fs=fs, n_src=args.n_speakers)
        return torch.from_numpy(mix), torch.from_numpy(tgt)
    return n, get, [args.samples] * n


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("ckpt", nargs="?", default=None)
    ap.add_argument("--synthetic-weights", type=int, default=0, metavar="NF")
    ap.add_argument("--dataset-dir", type=str, default=None)
    ap.add_argument("--synthetic", type=int, default=0, help="number of synthetic mixtures")
    ap.add_argument("--samples", type=int, default=32000)
    ap.add_argument("--n-speakers", type=int, default=2)
    ap.add_argument("-l", "--limit", type=int, default=None)
    ap.add_argument("-s", "--split", default="test", choices=["train", "val", "test", "libri2mix_test"])
    ap.add_argument("--cut", default="max", choices=["min", "max"])
    ap.add_argument("-N", type=int, default=None)
    ap.add_argument("--snr", type=float, default=None)
    ap.add_argument("--corrector-steps", type=int, default=None)
    ap.add_argument("--schedule", type=str, default=None)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("-o", "--output-dir", type=Path, default=Path("results"))
    ap.add_argument("--save-wav", action="store_true")
    ap.add_argument("--seed", type=int, default=0, help="torch.manual_seed before the first utterance: the i-th "
                                                         "utterance gets the i-th draw as its device RNG seed")
    ap.add_argument("--balance", action="store_true",
                    help="multi-GPU: deal the utterances to the ranks by length (longest first, round-robin) instead of "
                         "the reference's contiguous index ranges; needs the lengths (wav headers)")
    ap.add_argument("--streams", type=int, default=1,
                    help="utterances in flight per GPU: K engines on K HIP streams (results are bit-identical for any "
                         "K).  One utterance at a time (the reference's evaluation loop) leaves most of the GPU idle: "
                         "measured 6.0 utt/s with 1 stream, 10.4 / 13.5 / 18.5 with 2 / 3 / 4 (4 s utterances, "
                         "nf=64); more than 4 is slower again (streams share hardware queues).  Per-utterance "
                         "'runtime' is then the latency of an utterance that shared the GPU with K-1 others.")
    ap.add_argument("--enhance", action="store_true",
                    help="speech enhancement (evaluate.py:173-176,268-271): PriorMixSDE model, metrics on the first "
                         "source (clean speech) only")
    args = ap.parse_args(argv)
    if args.streams > 1:
        # HIP maps streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues, one of which the null stream holds:
        # with the default, two of four worker streams share a queue (measured 10.7 instead of 18.5 utt/s).  Read
        # by the HIP runtime when it initialises, i.e. this must precede the first torch.cuda call.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("No GPU visible: this build has no CPU path")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    def make_model():
        if args.synthetic_weights or args.ckpt is None:
            cfg = (enhancement_config(nf=args.synthetic_weights or 128) if args.enhance
                   else default_config(nf=args.synthetic_weights or 64, n_speakers=args.n_speakers))
            return DiffSepModel(cfg, dtype=args.dtype)
        return DiffSepModel.load_from_checkpoint(args.ckpt, dtype=args.dtype)

    K = max(1, args.streams)
    models = [make_model() for _ in range(K)]  # one engine (weights copy + workspace) per stream
    for m in models:
        # engines are created BEFORE the worker streams: HIP hands out hardware queues in stream-creation order, and
        # engines created lazily in between left the workers sharing queues (measured 7.0 instead of 17 utt/s, K=4)
        m.score_model.engine()
    model = models[0]
    fs = cfg_get(model.config, "model.fs", 8000)
    N = cfg_get(model.config, "model.sampler.N", 30) if args.N is None else args.N
    cs = cfg_get(model.config, "model.sampler.corrector_steps", 1) if args.corrector_steps is None else args.corrector_steps
    snr = cfg_get(model.config, "model.sampler.snr", 0.5) if args.snr is None else args.snr

    n, get, lengths = load_dataset(args, fs)  # lengths: samples per utterance when the wav headers tell (else None)
    max_len = max(lengths) if lengths else None
    if max_len:
        for m in models:  # workspace for the longest utterance now: growing it later would stall every stream
            m.score_model.engine().reserve(1, max_len)
    streams = [torch.cuda.Stream() for _ in range(K)]
    # the reference's contiguous ranges (evaluate_mp.py:495-503), or sorted by length and dealt round-robin
    mine = rank_indices(n, world, rank, lengths, args.balance)
    lo = mine[0] if mine else 0
    # warm every worker up on the first utterance's shape (engine creation, workspace plan, graph capture), then fix
    # the RNG state: results do not depend on the number of streams
    if mine:
        m0, _ = get(lo)
        for w in range(K):
            with torch.cuda.stream(streams[w]):
                mw = m0[None].cuda()
                (mw_n, _), *_ = models[w].normalize_batch((mw, None))
                models[w].get_pc_sampler("reverse_diffusion", "ald2", mw_n, N=N, corrector_steps=cs, snr=snr, denoise=True,
                                         intermediate=False, schedule=args.schedule)()
        torch.cuda.synchronize()
    # utterance i of the data set gets the i-th draw of a generator seeded with --seed as its device RNG seed: the
    # records do not depend on the number of streams, of ranks, or on how the utterances are dealt to them
    seeds = torch.randint(0, 2 ** 62, (max(n, 1),), generator=torch.Generator().manual_seed(args.seed)).tolist()
    records = []
    pending = [None] * K  # per worker: the utterance whose sampler is running on its stream

    def finish(w):
        if pending[w] is None:
            return
        i, mix, tgt_n, est, nfe, t0, _alive = pending[w]
        pending[w] = None
        streams[w].synchronize()
        runtime = time.perf_counter() - t0
        with torch.cuda.stream(streams[w]):
            if args.enhance:  # n_src = 1: only the clean-speech estimate is scored (evaluate.py:270)
                met = compute_metrics(est[0, :1], tgt_n[0, :1])
            else:
                met = compute_metrics(est[0], tgt_n[0])
        records.append({"batch_idx": i, **met, "pesq": None, "stoi": None, "nfe": int(nfe), "runtime": runtime,
                        "len_s": mix.shape[-1] / fs})
        if args.save_wav:
            d = args.output_dir / "wav"
            d.mkdir(parents=True, exist_ok=True)
            for k in range(est.shape[1]):
                wavio.save(d / f"{i:05d}_s{k}.wav", est[0, k:k + 1].cpu() * 0.1, fs)

    # One host thread drives all K streams (a thread per stream was measured SLOWER: 10.7 instead of 17 utt/s at K = 4;
    # concurrent launches serialise inside the HIP runtime and a launch that waits for queue space holds them all up).
    t_all = time.perf_counter()
    for j, i in enumerate(mine):
        w = j % K
        finish(w)  # the worker's previous utterance (oldest in flight)
        mix, tgt = get(i)
        with torch.cuda.stream(streams[w]):
            # pinned staging + asynchronous copies: a pageable host->device copy serialises the whole device
            mix = mix[None].contiguous().pin_memory().to("cuda", non_blocking=True)
            tgt = tgt[None].contiguous().pin_memory().to("cuda", non_blocking=True)
            (mix_n, tgt_n), *_ = models[w].normalize_batch((mix, tgt))
            sampler = models[w].get_pc_sampler("reverse_diffusion", "ald2", mix_n, N=N, corrector_steps=cs, snr=snr,
                                               denoise=True, intermediate=False, schedule=args.schedule,
                                               seed=seeds[i])
            if K == 1:
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            est, nfe, *_ = sampler()  # enqueues the whole sampler on the worker's stream
        # (every tensor the asynchronous sampler reads stays referenced until the worker's stream has drained)
        pending[w] = (i, mix, tgt_n, est, nfe, t0, (mix_n, tgt, sampler))
    for w in range(K):
        finish(w)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t_all
    allrec = gather_objects(records)
    if rank == 0:
        flat = sorted([r for part in allrec for r in part], key=lambda r: r["batch_idx"])
        args.output_dir.mkdir(parents=True, exist_ok=True)
        with open(args.output_dir / f"{args.split}.json", "w") as f:
            json.dump(flat, f, indent=2)
        summary = datasets.summarize([{k: v for k, v in r.items() if k not in ("batch_idx", "perm")} for r in flat])
        tot_rt = sum(r["runtime"] for r in flat)
        summary.update({"rtf": tot_rt / max(sum(r["len_s"] for r in flat), 1e-9), "world_size": world,
                        "streams": K, "utt_per_s_rank0": len(mine) / max(wall, 1e-9)})
        with open(args.output_dir / f"{args.split}_summary.json", "w") as f:
            json.dump(summary, f, indent=2)
        print(json.dumps(summary))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
