"""evaluate.py — the sampler + timing part of the reference's evaluate.py / evaluate_mp.py
(evaluate.py:322-443, evaluate_mp.py:154-326,495-528) on the HIP engine, one rank per GPU.

    python -m diffsep_amd.evaluate --synthetic 32 --synthetic-weights 64 -o out/            (1 GPU)
    python -m torch.distributed.run --nproc-per-node 8 -m diffsep_amd.evaluate ...          (8 GPUs)

Utterances are sharded over ranks in contiguous ranges (evaluate_mp.py:495-503); each rank separates its
share, records {batch_idx, si_sdr, si_sir, si_sar, pesq, stoi, nfe, runtime, len_s} per utterance (evaluate.py:394-405;
runtime is measured WITH a device sync, unlike evaluate.py:374-376) and rank 0 gathers everything (RCCL) and writes
<split>.json + <split>_summary.json (evaluate.py:436-443).  Dataset: --dataset-dir ROOT in the WSJ0-mix layout
(datasets/wsj0_mix.py:64-92; with --enhance the VoiceBank-DEMAND layout, datasets/vctk_demand.py:33-36), a flat
ROOT/{mix,s1,s2} folder, or --synthetic N speech-like mixtures.  SI-SDR (scale-invariant
SDR with the best source permutation) is computed in the normalised domain like evaluate.py:360,382.

The reference separates one utterance per sampler call (batch_size=1, evaluate.py:328) because lengths differ.  Here
utterances whose padded spectrogram width W = 64 ceil(F / 64) is equal ride in ONE engine call (--batch, default 16):
the batch is zero-padded on the right to its longest member, the engine keeps every utterance's tail at exactly zero
(diffsep_sampler_ext.lengths_host) and draws its noise from the utterance's own seed, so an utterance's record does
not depend on which batch, stream or rank it was separated in (bit-for-bit with --dtype f32; to the rounding of the
GroupNorm sums of the weight-stationary bf16 convolution otherwise).
"""
import argparse
import json
import os
import time
from pathlib import Path

import torch
import torch.distributed as dist

from . import datasets, metrics, ops, synth, wavio
from .dist_utils import gather_objects, rank_indices
from .pl_model import DiffSepModel, cfg_get, default_config, enhancement_config


def compute_metrics(est, ref, n_src=None):
    """est, ref [B,S,T] (zero-padded batches are fine: the zero tails add nothing to the Gram sums) -> per utterance a
    dict with the reference's metric fields (evaluate.py:103-132): the FULL source set is scored with the best
    permutation, then the first n_src entries are kept (n_src = 1 with --enhance: the clean speech; its permutation
    against the noise channel is still searched, evaluate.py:105-111,125-127).  The waveform reductions run in the HIP
    Gram kernel."""
    sdr, sir, sar, perm = metrics.si_bss_eval_sources(ref, est)
    k = sdr.shape[1] if n_src is None else n_src
    return [{"si_sdr": [[float(v) for v in sdr[b, :k]]], "si_sir": [[float(v) for v in sir[b, :k]]],
             "si_sar": [[float(v) for v in sar[b, :k]]], "perm": [int(v) for v in perm[b]]}
            for b in range(sdr.shape[0])]


def load_dataset(args, fs):
    """-> (number of utterances, get(i) -> (mix [1,T], tgt [S,T]) CPU tensors, lengths in samples)"""
    if args.dataset_dir and args.enhance:
        ds = datasets.NoisyDataset(args.dataset_dir, fs=fs, split=args.split)
        n = len(ds) if args.limit is None else min(len(ds), args.limit)
        return n, (lambda i: ds[i]), [ds.num_samples(i) for i in range(n)]
    if args.dataset_dir:
        root = Path(args.dataset_dir)
        if (root / "mix").is_dir():  # flat folder: mix/, s1/, s2/ ...
            names = sorted(p.name for p in (root / "mix").glob("*.wav"))[: args.limit]
            ds = datasets.WavPairs(root / "mix", [root / f"s{k + 1}" for k in range(args.n_speakers)], names, fs)
        else:
            ds = datasets.WSJ0_mix(root, n_spkr=args.n_speakers, fs=fs, cut=args.cut, split=args.split,
                                   max_n_samples=args.limit)
        return len(ds), (lambda i: tuple(t[..., : ds.num_samples(i)] for t in ds[i])), \
            [ds.num_samples(i) for i in range(len(ds))]
    n = args.synthetic
    hi = args.samples if args.samples_max is None else max(args.samples, args.samples_max)
    lens = [args.samples + (i * 7919) % (hi - args.samples + 1) for i in range(n)]  # (--samples-max: varied lengths)

    def get(i):
        mix, tgt = synth.synth_mixture(i, T=lens[i], fs=fs, n_src=args.n_speakers)
        return torch.from_numpy(mix), torch.from_numpy(tgt)
    return n, get, lens


def plan_batches(indices, lengths, width_of, batch):
    """Group utterance indices into engine batches: equal padded width W, at most `batch` per call, longest first
    inside a width (deterministic: ties by index).  Returns a list of index lists."""
    by_w = {}
    for i in indices:
        by_w.setdefault(width_of(lengths[i]), []).append(i)
    out = []
    for w in sorted(by_w):
        g = sorted(by_w[w], key=lambda i: (-lengths[i], i))
        out += [g[k:k + batch] for k in range(0, len(g), batch)]
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("ckpt", nargs="?", default=None)
    ap.add_argument("--synthetic-weights", type=int, default=0, metavar="NF")
    ap.add_argument("--dataset-dir", type=str, default=None)
    ap.add_argument("--synthetic", type=int, default=0, help="number of synthetic mixtures")
    ap.add_argument("--samples", type=int, default=32000)
    ap.add_argument("--samples-max", type=int, default=None,
                    help="--synthetic: utterance lengths spread over [--samples, --samples-max] instead of one length")
    ap.add_argument("--n-speakers", type=int, default=2)
    ap.add_argument("-l", "--limit", type=int, default=None)
    ap.add_argument("-s", "--split", default="test", choices=["train", "val", "test", "libri2mix_test"])
    ap.add_argument("--cut", default="max", choices=["min", "max"])
    ap.add_argument("-N", type=int, default=None)
    ap.add_argument("--snr", type=float, default=None)
    ap.add_argument("--corrector-steps", type=int, default=None)
    ap.add_argument("--schedule", type=str, default=None)
    ap.add_argument("--dtype", default="auto", choices=["auto", "f16", "bf16", "f32", "split", "hybrid"],
                    help="auto (default): f16 for backbones up to nf = 64, hybrid for wider ones; f16: 16-bit tensors in IEEE half precision, 50 dB from the fp32 result after 60 network "
                         "evaluations; bf16: the same kernels on bfloat16 tensors (32 dB); split / f32: fp32 tensors (bf16x3 / "
                         "exact fp32 matrix products); hybrid: f16 with the first reverse steps on a split engine")
    ap.add_argument("-o", "--output-dir", type=Path, default=Path("results"))
    ap.add_argument("--save-wav", action="store_true")
    ap.add_argument("--seed", type=int, default=0, help="torch.manual_seed before the first utterance: the i-th "
                                                         "utterance gets the i-th draw as its device RNG seed")
    ap.add_argument("--balance", action="store_true",
                    help="multi-GPU: deal the utterances to the ranks by length (longest first, round-robin) instead of "
                         "the reference's contiguous index ranges; needs the lengths (wav headers)")
    ap.add_argument("--batch", type=int, default=16,
                    help="utterances per engine call: those with the same padded spectrogram width share a call "
                         "(zero-padded to the longest; --batch 1 = the reference's one-utterance loop)")
    ap.add_argument("--streams", type=int, default=4,
                    help="engine calls (batches) in flight per GPU: K engines on K HIP streams; the records do not "
                         "depend on K.  'runtime' of an utterance is its batch's latency / batch size.")
    ap.add_argument("--fp32-steps", type=int, default=None,
                    help="with --dtype hybrid: the first K reverse steps run on the fp32 engine (default: pl_model.HYBRID_HEAD_STEPS)")
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
            return DiffSepModel(cfg, dtype=args.dtype, head_steps=args.fp32_steps)
        return DiffSepModel.load_from_checkpoint(args.ckpt, dtype=args.dtype, head_steps=args.fp32_steps)

    K = max(1, args.streams)
    models = [make_model() for _ in range(K)]  # one engine (weights copy + workspace) per stream
    for m in models:
        # engines are created BEFORE the worker streams: HIP hands out hardware queues in stream-creation order, and
        # engines created lazily in between left the workers sharing queues (measured 7.0 instead of 17 utt/s, K=4)
        m.score_model.engine()
        if m.tail_engine() is not None:
            m.tail_engine()
    model = models[0]
    eng0 = model.score_model.engine()
    fs = cfg_get(model.config, "model.fs", 8000)
    N = cfg_get(model.config, "model.sampler.N", 30) if args.N is None else args.N
    cs = cfg_get(model.config, "model.sampler.corrector_steps", 1) if args.corrector_steps is None else args.corrector_steps
    snr = cfg_get(model.config, "model.sampler.snr", 0.5) if args.snr is None else args.snr
    n_src = 1 if args.enhance else None  # (evaluate.py:268-271)

    n, get, lengths = load_dataset(args, fs)
    # the reference's contiguous ranges (evaluate_mp.py:495-503), or sorted by length and dealt round-robin
    mine = rank_indices(n, world, rank, lengths, args.balance)
    batches = plan_batches(mine, lengths, eng0.padded_frames, max(1, args.batch))
    if batches:  # workspace for the largest call now: growing it later would stall every stream
        bmax = max(len(g) for g in batches)
        tmax = eng0.bucket_length(eng0.padded_frames(max(lengths[i] for i in mine)))
        for m in models:
            m.score_model.engine().reserve(bmax, tmax)
            if m.tail_engine() is not None:
                m.tail_engine().reserve(bmax, tmax)
    streams = [torch.cuda.Stream() for _ in range(K)]
    # utterance i of the data set gets the i-th draw of a generator seeded with --seed as its device RNG seed: the
    # records do not depend on the number of streams, of ranks, or on how the utterances are batched or dealt
    seeds = torch.randint(0, 2 ** 62, (max(n, 1),), generator=torch.Generator().manual_seed(args.seed)).tolist()
    records = []
    fallbacks = []  # batches repeated on the split-precision engine after non-finite f16 samples
    pending = [None] * K  # per worker: the batch whose sampler is running on its stream

    def host_stage(group):
        """load and pad one batch on the host (runs on a loader thread, ahead of the GPU): mix / tgt + lengths"""
        items = [get(i) for i in group]
        # padded to the longest length of the batch's width bucket: one workspace plan / captured graph per (B, W)
        mix, tgt, lens = datasets.pad_batch(items, side="right",
                                            to=eng0.bucket_length(eng0.padded_frames(max(lengths[i] for i in group))))
        return mix.contiguous(), tgt.contiguous(), lens

    # the reference's DataLoader has worker processes; here two loader threads read / synthesise and pad the next batches
    # while the GPU separates the current ones (wav decoding and numpy release the GIL)
    from concurrent.futures import ThreadPoolExecutor
    loader = ThreadPoolExecutor(max_workers=2)
    ahead = {}

    def prefetch(j):
        for jj in range(j, min(j + 2 * K + 2, len(batches))):
            if jj not in ahead:
                ahead[jj] = loader.submit(host_stage, batches[jj])

    def stage(group, w, j=None):
        """upload and normalise one batch on worker w's stream -> (mix, mix_n, tgt_n, lens)"""
        mix, tgt, lens = ahead.pop(j).result() if j in ahead else host_stage(group)
        # pinned staging + asynchronous copies (a pageable host->device copy serialises the whole device); pinned memory is
        # allocated on THIS thread: the loader threads make no HIP runtime call
        mix = mix.pin_memory().to("cuda", non_blocking=True)
        tgt = tgt.pin_memory().to("cuda", non_blocking=True)
        mix_n, tgt_n = torch.zeros_like(mix), torch.zeros_like(tgt)
        for b, L in enumerate(lens):  # every utterance is normalised over ITS samples (pl_model.py:81-88)
            (m_b, t_b), *_ = models[w].normalize_batch((mix[b:b + 1, :, :L], tgt[b:b + 1, :, :L]))
            mix_n[b, :, :L], tgt_n[b, :, :L] = m_b[0], t_b[0]
        return mix, mix_n, tgt_n, lens

    def launch(group, w, j=None):
        mix, mix_n, tgt_n, lens = stage(group, w, j)
        sampler = models[w].get_pc_sampler("reverse_diffusion", "ald2", mix_n, N=N, corrector_steps=cs, snr=snr,
                                           denoise=True, intermediate=False, schedule=args.schedule,
                                           lengths=lens, seeds=[seeds[i] for i in group], check_finite=False)
        if K == 1:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        est, nfe, *_ = sampler()  # enqueues the whole sampler on the worker's stream
        # (every tensor the asynchronous sampler reads stays referenced until the worker's stream has drained)
        return (group, lens, tgt_n, est, nfe, t0, (mix, mix_n, sampler))

    def finish(w):
        if pending[w] is None:
            return
        group, lens, tgt_n, est, nfe, t0, _alive = pending[w]
        pending[w] = None
        streams[w].synchronize()
        # half precision overflows at 65504: a batch with non-finite samples is repeated on the model's split-precision twin
        # (DiffSepModel.rerun_if_nonfinite — the one place that decides; raises if that is non-finite too)
        def rerun(fb):
            with torch.cuda.stream(streams[w]):
                r = fb.get_pc_sampler("reverse_diffusion", "ald2", _alive[1], N=N, corrector_steps=cs, snr=snr, denoise=True,
                                      intermediate=False, schedule=args.schedule, lengths=lens,
                                      seeds=[seeds[i] for i in group], check_finite=False)()
            streams[w].synchronize()
            fallbacks.append(list(group))
            return r
        est, nfe, *_ = models[w].rerun_if_nonfinite((est, nfe), rerun, what=f"utterances {group[:3]}...")
        runtime = (time.perf_counter() - t0) / len(group)
        with torch.cuda.stream(streams[w]):
            mets = compute_metrics(est, tgt_n, n_src)
        for b, i in enumerate(group):
            records.append({"batch_idx": i, **mets[b], "pesq": None, "stoi": None, "nfe": int(nfe),
                            "runtime": runtime, "len_s": lens[b] / fs})
            if args.save_wav:
                d = args.output_dir / "wav"
                d.mkdir(parents=True, exist_ok=True)
                for k in range(est.shape[1]):
                    wavio.save(d / f"{i:05d}_s{k}.wav", est[b, k:k + 1, :lens[b]].cpu() * 0.1, fs, bits=32)

    # warm every worker up on the first batch's shape (workspace plan, graph capture) outside the timed region
    if batches:
        for w in range(K):
            with torch.cuda.stream(streams[w]):
                launch(batches[0], w)
        torch.cuda.synchronize()
    # One host thread drives all K streams (a thread per stream was measured SLOWER: 10.7 instead of 17 utt/s at K = 4;
    # concurrent launches serialise inside the HIP runtime and a launch that waits for queue space holds them all up).
    t_all = time.perf_counter()
    prefetch(0)
    for j, group in enumerate(batches):
        w = j % K
        finish(w)  # the worker's previous batch (oldest in flight)
        prefetch(j)
        with torch.cuda.stream(streams[w]):
            pending[w] = launch(group, w, j)
    for w in range(K):
        finish(w)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t_all
    loader.shutdown()
    allrec = gather_objects(records)
    if rank == 0:
        flat = sorted([r for part in allrec for r in part], key=lambda r: r["batch_idx"])
        args.output_dir.mkdir(parents=True, exist_ok=True)
        with open(args.output_dir / f"{args.split}.json", "w") as f:
            json.dump(flat, f, indent=2)
        summary = datasets.summarize([{k: v for k, v in r.items() if k not in ("batch_idx", "perm")} for r in flat])
        tot_rt = sum(r["runtime"] for r in flat)
        summary.update({"rtf": tot_rt / max(sum(r["len_s"] for r in flat), 1e-9), "world_size": world,
                        "streams": K, "batch": args.batch, "engine_calls_rank0": len(batches), "dtype": model.dtype,
                        "utt_per_s_rank0": len(mine) / max(wall, 1e-9), "split_fallback_batches_rank0": len(fallbacks),
                        # metrics this build does not compute (third-party C code, out of scope: DESIGN.md section 7)
                        "not_computed": ["pesq", "stoi"]})
        with open(args.output_dir / f"{args.split}_summary.json", "w") as f:
            json.dump(summary, f, indent=2)
        print(json.dumps(summary))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
