"""evaluate.py — the reference's evaluate.py / evaluate_mp.py (evaluate.py:164-443, evaluate_mp.py:154-326,495-528) on the
HIP engine, one rank per GPU.  The reference's command line runs verbatim:

    python -m diffsep_amd.evaluate exp/default/2022-xx/checkpoints/epoch-979.ckpt --test -s log -d 1 --save-n 10 -o results
    python -m diffsep_amd.evaluate --synthetic 32 --synthetic-weights 64 -o out/            (no data set / checkpoint at hand)
    python -m torch.distributed.run --nproc-per-node 8 -m diffsep_amd.evaluate ...          (8 GPUs)

Reference flags (evaluate.py:168-226), same names, defaults and meaning: ckpt (or `__no_proc__`: score the unprocessed
mixture, :245-262), -o/--output_dir, --enhance, -d/--device, -w/--dl-workers, --tag, -l/--limit, --save-n, --val, --test
(one of the two is required unless an extension below names the data), -N, --snr, --corrector-steps, --denoise (argparse
`type=bool` as in the reference: every non-empty value is True), --pesq-mode, --stoi-no-extended, -s/--schedule.  The data
sets come from `<ckpt>/../../hparams.yaml` -> datamodule.{val,test}.dataset like evaluate.py:264-288 (or --dataset-dir).
Output tree (evaluate.py:306-323,436-443): `<output_dir>/<exp>_<ckpt>_<tag_inf>/` (or `<tag>_<tag_inf>`), tag_inf =
`N-.._snr-.._corrstep-.._denoise-.._schedule-..`, holding `<split>.json`, `<split>_summary.json` and
`wav/<split>/NNN_{mix,enh0,enh1,tgt0,tgt1}.wav` (scaled to peak 0.95 together, estimates in the permutation that matches
the targets) for the first --save-n utterances (default: all).  Not produced: `fig/` (matplotlib spectrogram plots of the
intermediate states) and the PESQ field (ITU-T P.862 C code, third-party `pesq`; null, named under `not_computed`).  STOI /
ESTOI is computed (diffsep_amd.metrics.stoi) on loader threads.

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


def _hparams_datasets(args, fs_model, splits):
    """evaluate.py:264-288: the data sets named by the experiment's hparams.yaml (two levels above the checkpoint)."""
    import yaml
    hp = Path(args.ckpt).parents[1] / "hparams.yaml"
    if not hp.exists():
        raise SystemExit(f"{hp} not found: the reference reads the data set from it (evaluate.py:265-267); pass --dataset-dir "
                         "ROOT or --synthetic N instead")
    with open(hp, "r") as f:
        config = yaml.safe_load(f)["config"]
    out = {}
    if args.enhance:
        kw = dict(config["datamodule"]["test"]["dataset"])
        kw.pop("_target_", None)
        out["test"] = datasets.NoisyDataset(**kw)
        return out
    for split in splits:
        kw = dict(config["datamodule"][split]["dataset"])
        kw.pop("_target_", None)
        if not Path(kw["path"]).exists():
            kw["path"] = "./data/wsj0_mix"
        out[split] = datasets.WSJ0_mix(**kw)
    return out


def save_samples(mix, est, tgt, wav_out_dir, idx, fs):
    """evaluate.py:70-101: mixture, estimates (already in the targets' order) and targets, scaled TOGETHER to peak 0.95, as
    32-bit float wav (what torchaudio.save writes for a float tensor).  Every source is written (the reference's fixed five
    files are these for two sources)."""
    allw = torch.cat((mix, est, tgt), dim=0).clone()
    allw *= 0.95 / allw.abs().max().clamp(min=1e-30)
    S = est.shape[0]
    wav_out_dir.mkdir(parents=True, exist_ok=True)
    wavio.save(wav_out_dir / f"{idx:03d}_mix.wav", allw[0:1], fs, bits=32)
    for k in range(S):
        wavio.save(wav_out_dir / f"{idx:03d}_enh{k}.wav", allw[1 + k:2 + k], fs, bits=32)
    for k in range(tgt.shape[0]):
        wavio.save(wav_out_dir / f"{idx:03d}_tgt{k}.wav", allw[1 + S + k:2 + S + k], fs, bits=32)


def build_parser():
    ap = argparse.ArgumentParser(description="Run evaluation on validation or test dataset")
    # ---- the reference's arguments (evaluate.py:168-226)
    ap.add_argument("ckpt", nargs="?", default=None, type=Path, help="Path to checkpoint to use ('__no_proc__': score the mixture)")
    ap.add_argument("-o", "--output_dir", "--output-dir", dest="output_dir", type=Path, default=Path("results"), help="The output folder")
    ap.add_argument("--enhance", default=False, action="store_true",
                    help="Compute evaluation metrics for speech enhancement (evaluate.py:173-176,268-271): PriorMixSDE model, "
                         "metrics on the first source (clean speech) only")
    ap.add_argument("-d", "--device", default=0, help="Device to use (default: cuda:0); under torchrun LOCAL_RANK decides")
    ap.add_argument("-w", "--dl-workers", type=int, default=None,
                    help="Number of loader / STOI worker threads (default min(os.cpu_count(), 16))")
    ap.add_argument("--tag", type=str, default=None,
                    help="A tag name for the experiment. If not provided, the experiment and checkpoints name are used.")
    ap.add_argument("-l", "--limit", type=int, default=None, help="Limit the number of samples to process")
    ap.add_argument("--save-n", type=int, default=None, help="Save a limited number of output samples (default: save all)")
    ap.add_argument("--val", action="store_true", help="Run on validation dataset")
    ap.add_argument("--test", action="store_true", help="Run on test dataset")
    ap.add_argument("-N", type=int, default=None, help="Number of steps")
    ap.add_argument("--snr", type=float, default=None, help="Step size of corrector")
    ap.add_argument("--corrector-steps", type=int, default=None, help="Number of corrector steps")
    ap.add_argument("--denoise", type=bool, default=True, help="Use denoising in solver")
    ap.add_argument("--pesq-mode", type=str, choices=["nb", "wb"], default="nb",
                    help="Mode for PESQ 'wb' or 'nb' (accepted; PESQ is not computed: see not_computed in the summary)")
    ap.add_argument("--stoi-no-extended", action="store_true", help="Disable extended mode for STOI")
    ap.add_argument("-s", "--schedule", type=str, default=None, help="Pick a different schedule for the inference")
    # ---- extensions
    ap.add_argument("--split", default=None, choices=["train", "val", "test", "libri2mix_test"],
                    help="with --dataset-dir: the split folder (default test); --val / --test select it the reference's way")
    ap.add_argument("--synthetic-weights", type=int, default=0, metavar="NF")
    ap.add_argument("--dataset-dir", type=str, default=None)
    ap.add_argument("--synthetic", type=int, default=0, help="number of synthetic mixtures")
    ap.add_argument("--samples", type=int, default=32000)
    ap.add_argument("--samples-max", type=int, default=None,
                    help="--synthetic: utterance lengths spread over [--samples, --samples-max] instead of one length")
    ap.add_argument("--n-speakers", type=int, default=2)
    ap.add_argument("--cut", default="max", choices=["min", "max"])
    ap.add_argument("--dtype", default="auto", choices=["auto", "f16", "bf16", "f32", "split", "hybrid"],
                    help="auto (default): f16 for backbones up to nf = 64, hybrid for wider ones; f16: 16-bit tensors in IEEE half precision, 50 dB from the fp32 result after 60 network "
                         "evaluations; bf16: the same kernels on bfloat16 tensors (32 dB); split / f32: fp32 tensors (bf16x3 / "
                         "exact fp32 matrix products); hybrid: f16 with the first reverse steps on a split engine")
    ap.add_argument("--flat-output", action="store_true",
                    help="write <split>.json / <split>_summary.json / wav/ directly into --output_dir instead of the reference's "
                         "<output_dir>/<exp>_<ckpt>_<tag_inf>/ folder")
    ap.add_argument("--no-stoi", action="store_true", help="skip STOI (host-side, ~0.1 s per utterance and source on one core)")
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
    return ap


def main(argv=None):
    """Returns the folder the results were written to (rank 0; the reference's naming, evaluate.py:306-323)."""
    ap = build_parser()
    args = ap.parse_args(argv)
    splits = [s for s, on in (("val", args.val), ("test", args.test)) if on]
    if not splits:
        if args.split is not None or args.synthetic or args.dataset_dir:
            splits = [args.split or "test"]  # (extensions that name the data themselves)
        else:
            ap.error("No action requested, add --val or --test")
    if args.enhance:
        splits = ["test"] if not args.dataset_dir or args.split is None else [args.split]  # (evaluate.py:268-271: the test set)
    no_proc = str(args.ckpt) == "__no_proc__"
    if args.ckpt is None and not args.synthetic_weights and not no_proc:
        ap.error("a checkpoint (or --synthetic-weights NF) is required")
    if args.streams > 1:
        # HIP maps streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues, one of which the null stream holds:
        # with the default, two of four worker streams share a queue (measured 10.7 instead of 18.5 utt/s).  Read
        # by the HIP runtime when it initialises, i.e. this must precede the first torch.cuda call.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if "LOCAL_RANK" in os.environ:
        local = int(os.environ["LOCAL_RANK"])
    else:  # -d 1 / -d cuda:1 (evaluate.py:177-179)
        d = str(args.device)
        local = int(d.split(":")[1]) if ":" in d else (int(d) if d.isdigit() else 0)
    if not torch.cuda.is_available():
        raise SystemExit("No GPU visible: this build has no CPU path")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    n_workers = max(2, min(os.cpu_count() or 2, 16) if args.dl_workers is None else args.dl_workers)

    K = max(1, args.streams)
    models, model = [], None
    if not no_proc:
        if args.synthetic_weights or args.ckpt is None:
            cfg = (enhancement_config(nf=args.synthetic_weights or 128) if args.enhance
                   else default_config(nf=args.synthetic_weights or 64, n_speakers=args.n_speakers))
            model = DiffSepModel(cfg, dtype=args.dtype, head_steps=args.fp32_steps)
        else:
            model = DiffSepModel.load_from_checkpoint(args.ckpt, dtype=args.dtype, head_steps=args.fp32_steps)
        model.eval()
        # one engine (weights repacked on the device + workspace) per stream over ONE set of parameters
        models = [model] + [model.replica() for _ in range(K - 1)]
        for m in models:
            if K > 1:
                m.set_throughput_mode(True)
            # engines are created BEFORE the worker streams: HIP hands out hardware queues in stream-creation order, and
            # engines created lazily in between left the workers sharing queues (measured 7.0 instead of 17 utt/s, K=4)
            m.score_model.engine()
            if m.tail_engine() is not None:
                m.tail_engine()
    fs_model = cfg_get(model.config, "model.fs", 8000) if model is not None else None
    N = cs = snr = None
    if model is not None:
        N = cfg_get(model.config, "model.sampler.N", 30) if args.N is None else args.N
        cs = cfg_get(model.config, "model.sampler.corrector_steps", 1) if args.corrector_steps is None else args.corrector_steps
        snr = cfg_get(model.config, "model.sampler.snr", 0.5) if args.snr is None else args.snr
    denoise = args.denoise
    n_src = 1 if args.enhance else None  # (evaluate.py:268-271)

    # ---- the output folder (evaluate.py:257-262,306-323)
    if args.flat_output:
        output_dir = args.output_dir
    elif no_proc:
        output_dir = args.output_dir / ("mix" if args.tag is None else args.tag)
    else:
        tag_inf = f"N-{N}_snr-{snr}_corrstep-{cs}_denoise-{denoise}_schedule-{args.schedule}"
        if args.tag is not None:
            output_dir = args.output_dir / f"{args.tag}_{tag_inf}"
        elif args.ckpt is not None and not args.synthetic_weights:
            output_dir = args.output_dir / f"{Path(args.ckpt).absolute().parents[1].name}_{Path(args.ckpt).stem}_{tag_inf}"
        else:
            output_dir = args.output_dir / f"synthetic-nf{args.synthetic_weights}_random-init_{tag_inf}"
    if rank == 0:
        output_dir.mkdir(exist_ok=True, parents=True)
        print(f"Created output folder {output_dir}")

    # ---- the data sets, one per split (evaluate.py:245-288)
    from_hparams = None
    if not (args.synthetic or args.dataset_dir):
        if no_proc:  # evaluate.py:247-255
            from_hparams = {s: datasets.WSJ0_mix(path="data/wsj0_mix", n_spkr=2, cut="max", split=s) for s in splits}
        else:
            from_hparams = _hparams_datasets(args, fs_model, splits)

    streams = [torch.cuda.Stream() for _ in range(K)]
    from concurrent.futures import ThreadPoolExecutor
    # the reference's DataLoader has worker processes; here loader threads read / synthesise and pad the next batches while the
    # GPU separates the current ones (wav decoding and numpy release the GIL), and score STOI of the finished ones
    loader = ThreadPoolExecutor(max_workers=n_workers)
    for split in splits:
        if from_hparams is not None:
            ds = from_hparams[split]
            n_ = len(ds) if args.limit is None else min(len(ds), args.limit)
            data = (n_, (lambda i, ds=ds: tuple(t[..., : ds.num_samples(i)] for t in ds[i])), [ds.num_samples(i) for i in range(n_)])
            fs = ds.fs
        else:
            args_split = argparse.Namespace(**{**vars(args), "split": split})
            fs = fs_model if fs_model is not None else 8000
            data = load_dataset(args_split, fs)
        run_split(args, split, data, fs, models, streams, loader, output_dir, world, rank, N, cs, snr, denoise, n_src, no_proc)
    loader.shutdown()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return output_dir


def run_split(args, split, data, fs, models, streams, loader, output_dir, world, rank, N, cs, snr, denoise, n_src, no_proc):
    n, get, lengths = data
    K = len(streams)
    model = models[0] if models else None
    if rank == 0:
        print(f"Processing {split}: {n} samples")
    if no_proc:
        width_of = lambda T: 64 * ((1 + (T + 382) // 128 + 63) // 64)
        bucket = lambda W: 128 * W - 383
    else:
        eng0 = model.score_model.engine()
        width_of, bucket = eng0.padded_frames, eng0.bucket_length
    # the reference's contiguous ranges (evaluate_mp.py:495-503), or sorted by length and dealt round-robin
    mine = rank_indices(n, world, rank, lengths, args.balance)
    batches = plan_batches(mine, lengths, width_of, max(1, args.batch))
    if batches and not no_proc:  # workspace for the largest call now: growing it later would stall every stream
        bmax = max(len(g) for g in batches)
        tmax = bucket(width_of(max(lengths[i] for i in mine)))
        for m in models:
            m.score_model.engine().reserve(bmax, tmax)
            if m.tail_engine() is not None:
                m.tail_engine().reserve(bmax, tmax)
    # utterance i of the data set gets the i-th draw of a generator seeded with --seed as its device RNG seed: the
    # records do not depend on the number of streams, of ranks, or on how the utterances are batched or dealt
    seeds = torch.randint(0, 2 ** 62, (max(n, 1),), generator=torch.Generator().manual_seed(args.seed)).tolist()
    records = []
    fallbacks = []  # batches repeated on the split-precision engine after non-finite f16 samples
    pending = [None] * K  # per worker: the batch whose sampler is running on its stream
    stoi_jobs = []  # (record, future of the per-source STOI list)

    def host_stage(group):
        """load and pad one batch on the host (runs on a loader thread, ahead of the GPU): mix / tgt + lengths"""
        items = [get(i) for i in group]
        # padded to the longest length of the batch's width bucket: one workspace plan / captured graph per (B, W)
        mix, tgt, lens = datasets.pad_batch(items, side="right", to=bucket(width_of(max(lengths[i] for i in group))))
        return mix.contiguous(), tgt.contiguous(), lens

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
        if no_proc:  # (evaluate.py:349-355: the raw mixture against the raw targets)
            return mix, mix, tgt, lens
        mix_n, tgt_n = torch.zeros_like(mix), torch.zeros_like(tgt)
        for b, L in enumerate(lens):  # every utterance is normalised over ITS samples (pl_model.py:81-88)
            (m_b, t_b), *_ = models[w].normalize_batch((mix[b:b + 1, :, :L], tgt[b:b + 1, :, :L]))
            mix_n[b, :, :L], tgt_n[b, :, :L] = m_b[0], t_b[0]
        return mix, mix_n, tgt_n, lens

    def launch(group, w, j=None):
        mix, mix_n, tgt_n, lens = stage(group, w, j)
        if no_proc:
            est = mix_n.expand(-1, tgt_n.shape[1], -1).contiguous()  # x_result = broadcast_to(mix, target.shape)
            return (group, lens, tgt_n, est, 0, None, (mix, mix_n, None))
        sampler = models[w].get_pc_sampler("reverse_diffusion", "ald2", mix_n, N=N, corrector_steps=cs, snr=snr,
                                           denoise=denoise, intermediate=False, schedule=args.schedule,
                                           lengths=lens, seeds=[seeds[i] for i in group], check_finite=False)
        if K == 1:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        est, nfe, *_ = sampler()  # enqueues the whole sampler on the worker's stream
        # (every tensor the asynchronous sampler reads stays referenced until the worker's stream has drained)
        return (group, lens, tgt_n, est, nfe, t0, (mix, mix_n, sampler))

    def stoi_of(tgt_rows, est_rows):
        return [metrics.stoi(t_, e_, fs, extended=not args.stoi_no_extended) for t_, e_ in zip(tgt_rows, est_rows)]

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
                r = fb.get_pc_sampler("reverse_diffusion", "ald2", _alive[1], N=N, corrector_steps=cs, snr=snr, denoise=denoise,
                                      intermediate=False, schedule=args.schedule, lengths=lens,
                                      seeds=[seeds[i] for i in group], check_finite=False)()
            streams[w].synchronize()
            fallbacks.append(list(group))
            return r
        if not no_proc:
            est, nfe, *_ = models[w].rerun_if_nonfinite((est, nfe), rerun, what=f"utterances {group[:3]}...")
        runtime = 0.0 if t0 is None else (time.perf_counter() - t0) / len(group)
        with torch.cuda.stream(streams[w]):
            mets = compute_metrics(est, tgt_n, n_src)
        need_host = (not args.no_stoi) or args.save_n is None or any(i < args.save_n for i in group)
        est_h = tgt_h = mix_h = None
        if need_host:
            with torch.cuda.stream(streams[w]):
                est_h, tgt_h, mix_h = est.cpu(), tgt_n.cpu(), _alive[1].cpu()
        for b, i in enumerate(group):
            rec = {"batch_idx": i, **mets[b], "pesq": None, "stoi": None, "nfe": int(nfe), "runtime": runtime,
                   "len_s": lens[b] / fs}
            records.append(rec)
            perm = mets[b]["perm"]
            k_src = len(perm) if n_src is None else n_src
            if need_host:
                est_b = est_h[b, perm, :lens[b]]  # "fix the permutation" (evaluate.py:392): estimates in the targets' order
                if not args.no_stoi:
                    stoi_jobs.append((rec, loader.submit(stoi_of, tgt_h[b, :k_src, :lens[b]].numpy(), est_b[:k_src].numpy())))
                if args.save_n is None or i < args.save_n:  # (evaluate.py:341; figures are not produced)
                    save_samples(mix_h[b, :, :lens[b]], est_b, tgt_h[b, :, :lens[b]], output_dir / "wav" / split, i, fs)

    # warm every worker up on the first batch's shape (workspace plan, graph capture) outside the timed region
    if batches and not no_proc:
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
    for rec, fut in stoi_jobs:
        rec["stoi"] = fut.result()
    allrec = gather_objects(records)
    if rank == 0:
        flat = sorted([r for part in allrec for r in part], key=lambda r: r["batch_idx"])
        with open(output_dir / f"{split}.json", "w") as f:
            json.dump(flat, f, indent=2)
        summary = datasets.summarize([{k: v for k, v in r.items() if k not in ("batch_idx", "perm")} for r in flat])
        tot_rt = sum(r["runtime"] for r in flat)
        summary.update({"rtf": tot_rt / max(sum(r["len_s"] for r in flat), 1e-9), "world_size": world,
                        "streams": K, "batch": args.batch, "engine_calls_rank0": len(batches),
                        "dtype": model.dtype if model is not None else None,
                        "utt_per_s_rank0": len(mine) / max(wall, 1e-9), "split_fallback_batches_rank0": len(fallbacks),
                        # PESQ is ITU-T P.862 reference C code behind the third-party `pesq` package: not restated here
                        # (DESIGN.md section 7); STOI / ESTOI is diffsep_amd.metrics.stoi (published algorithm, restated)
                        "not_computed": ["pesq"] + (["stoi"] if args.no_stoi else []),
                        "stoi_extended": not args.stoi_no_extended, "pesq_mode": args.pesq_mode})
        with open(output_dir / f"{split}_summary.json", "w") as f:
            json.dump(summary, f, indent=2)
        print(json.dumps(summary))


if __name__ == "__main__":
    main()
