"""separate.py — same command line as the reference's separate.py:102-162:

    python -m diffsep_amd.separate input_dir output_dir [--model CKPT] [-d cuda:0] [-N 30] [--snr 0.5]
           [--corrector-steps 1] [--denoise True] [-s SCHEDULE]

For every *.wav in input_dir: load -> normalize_batch -> reverse-diffusion PC sampler on the HIP engine ->
scale_output -> write output_dir/s{i}/name.wav (directories s0, s1 like separate.py:157-158).
Additions: --synthetic-weights NF runs with random-init weights of width NF when no checkpoint is
available (there is no network here: the HF default 'fakufaku/diffsep' cannot be downloaded), --dtype (bf16 / f32 /
hybrid), --batch B: files whose padded spectrogram width is equal share one engine call (zero-padded to the longest,
each file's tail kept at zero by the engine: evaluate.py has the details), --streams K engine calls in flight on K
engines / HIP streams, and --seed (file i of the sorted folder then gets the i-th draw of that generator as its device
RNG seed: the written files do not depend on --batch or --streams).  Output files are 32-bit float WAV like
torchaudio.save of a float tensor (separate.py:160-162).
"""
import argparse
import os
from pathlib import Path

import torch

from . import datasets, ops, wavio
from .pl_model import DiffSepModel, cfg_get, default_config

DEFAULT_MODEL = "fakufaku/diffsep"


def get_model(args):
    if args.synthetic_weights:
        model = DiffSepModel(default_config(nf=args.synthetic_weights), dtype=args.dtype, device=args.device,
                             head_steps=getattr(args, "fp32_steps", None))
    else:
        path = Path(args.model)
        if not path.exists():
            raise FileNotFoundError(f"checkpoint '{args.model}' not found (Hugging Face download needs network access; "
                                    "pass a local Lightning checkpoint or --synthetic-weights NF)")
        model = DiffSepModel.load_from_checkpoint(str(path), dtype=args.dtype, device=args.device,
                                                  head_steps=getattr(args, "fp32_steps", None))
    model.to(args.device)
    model.eval()
    N = cfg_get(model.config, "model.sampler.N", 30) if args.N is None else args.N
    cs = cfg_get(model.config, "model.sampler.corrector_steps", 1) if args.corrector_steps is None else args.corrector_steps
    snr = cfg_get(model.config, "model.sampler.snr", 0.5) if args.snr is None else args.snr
    kwargs = {"N": N, "denoise": args.denoise, "intermediate": False, "corrector_steps": cs, "snr": snr,
              "schedule": args.schedule}
    return model, kwargs


def scale_output(mix, sep):
    """separate.py:73-78, in the HIP kernel."""
    return ops.scale_output(mix.contiguous(), sep.contiguous())


def separate_on_device(mix, model, sampler_kwargs, device, lengths=None, seeds=None, check_finite=False):
    """Enqueue the separation of mix [1,T] / [B,1,T] on the current stream; returns the device tensor [B,S,T].
    lengths [B]: mix is a right-zero-padded batch of files of those lengths (normalised and rescaled per file)."""
    mix = mix.to(device)
    if mix.dim() == 2:
        mix = mix[None]
    if lengths is None:
        (mix_norm, _), *_ = model.normalize_batch((mix, None))
    else:
        mix_norm = torch.zeros_like(mix)
        for b, L in enumerate(lengths):
            mix_norm[b, :, :L] = model.normalize_batch((mix[b:b + 1, :, :L], None))[0][0][0]
    extra = {} if lengths is None else {"lengths": list(lengths)}
    if seeds is not None:
        extra["seeds"] = list(seeds)
    # (check_finite=False: the callers below collect the result later and run the model's overflow net then)
    sampler = model.get_pc_sampler("reverse_diffusion", "ald2", mix_norm, check_finite=check_finite, **sampler_kwargs, **extra)
    with torch.no_grad():
        sep, nfe, *_ = sampler()
    if lengths is None:
        return scale_output(mix, sep)
    out = torch.zeros_like(sep)
    for b, L in enumerate(lengths):  # the least-squares scale of separate.py:73-78 is per file, over ITS samples
        out[b, :, :L] = scale_output(mix[b:b + 1, :, :L], sep[b:b + 1, :, :L])[0]
    return out


def separate(mix, model, sampler_kwargs, device):
    """mix [1,T] (one file, like the reference) or [B,1,T] (a batch of equal-length files)."""
    return separate_on_device(mix, model, sampler_kwargs, device, check_finite=True).cpu()


def main(argv=None):
    ap = argparse.ArgumentParser(description="Separate all the wav files in a specified folder")
    ap.add_argument("input_dir", type=Path)
    ap.add_argument("output_dir", type=Path)
    ap.add_argument("--model", type=str, default=DEFAULT_MODEL, help="Path to a Lightning checkpoint")
    ap.add_argument("-d", "--device", type=str, default="cuda:0")
    ap.add_argument("-N", type=int, default=None, help="Number of steps")
    ap.add_argument("--snr", type=float, default=None, help="Step size of corrector")
    ap.add_argument("--corrector-steps", type=int, default=None)
    ap.add_argument("--denoise", type=bool, default=True)
    ap.add_argument("-s", "--schedule", type=str, default=None)
    ap.add_argument("--synthetic-weights", type=int, default=0, metavar="NF")
    ap.add_argument("--dtype", default="auto", choices=["auto", "f16", "bf16", "f32", "split", "hybrid"],
                    help="auto (default): f16 for backbones up to nf = 64, hybrid for wider ones; f16: 16-bit tensors in IEEE half precision, 50 dB from the fp32 result after 60 network "
                         "evaluations; bf16: the same kernels on bfloat16 tensors (32 dB); split / f32: fp32 tensors (bf16x3 / "
                         "exact fp32 matrix products); hybrid: f16 with the first reverse steps on a split engine")
    ap.add_argument("--fp32-steps", type=int, default=None, help="--dtype hybrid: the first K reverse steps run on the fp32 engine")
    ap.add_argument("--batch", type=int, default=1, help="files per engine call (equal padded width)")
    ap.add_argument("--streams", type=int, default=1, help="engine calls in flight: K engines on K HIP streams")
    ap.add_argument("--seed", type=int, default=None,
                    help="file i (sorted) gets the i-th draw of a generator with this seed as its device RNG seed")
    args = ap.parse_args(argv)
    K = max(1, args.streams)
    if K > 1:  # (see evaluate.py: hardware queues; must precede the first torch.cuda call)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if not torch.cuda.is_available():
        raise SystemExit("No GPU visible: this build has no CPU path (the reference falls back to CPU here)")
    torch.cuda.set_device(torch.device(args.device))
    model, kw = get_model(args)
    # one engine (weights repacked on the device + workspace) per stream over ONE set of parameters
    models = [model] + [model.replica() for _ in range(K - 1)]
    for m in models:
        if K > 1:
            m.set_throughput_mode(True)  # (several batches in flight: pl_model.DiffSepModel.set_throughput_mode)
        m.score_model.engine()  # engines before streams (hardware queues are handed out in creation order)
        if m.tail_engine() is not None:
            m.tail_engine()
    model_sr = cfg_get(model.config, "model.fs", 8000)
    if args.output_dir.is_file():
        raise ValueError("Output directory is a file")
    args.output_dir.mkdir(parents=True, exist_ok=True)
    files = sorted(args.input_dir.glob("*.wav"))
    lengths = [wavio.info(p)[1] for p in files]
    eng = model.score_model.engine()
    from .evaluate import plan_batches
    batches = plan_batches(range(len(files)), lengths, eng.padded_frames, max(1, args.batch))
    if batches:  # workspace for the largest call now (growing it later synchronises the whole device)
        for m in models:
            tmax = eng.bucket_length(eng.padded_frames(max(lengths)))
            m.score_model.engine().reserve(max(len(g) for g in batches), tmax)
            if m.tail_engine() is not None:
                m.tail_engine().reserve(max(len(g) for g in batches), tmax)
    seeds = None
    if args.seed is not None:
        seeds = torch.randint(0, 2 ** 62, (max(len(files), 1),),
                              generator=torch.Generator().manual_seed(args.seed)).tolist()
    streams = [torch.cuda.Stream() for _ in range(K)] if K > 1 else [torch.cuda.current_stream()]
    in_flight = [None] * K  # per worker: (file indices, lengths, sample rates, device result) of its running batch

    def finish(w):
        if in_flight[w] is None:
            return
        group, lens, srs, sep, mix_d, sds = in_flight[w]
        in_flight[w] = None
        streams[w].synchronize()
        # half precision overflows at 65504: non-finite samples -> the batch is repeated on the model's split-precision twin
        # (DiffSepModel.rerun_if_nonfinite, the one place that decides)
        def rerun(fb):
            with torch.cuda.stream(streams[w]):
                r = separate_on_device(mix_d, fb, kw, args.device, lengths=lens, seeds=sds)
            streams[w].synchronize()
            return (r,)
        sep = models[w].rerun_if_nonfinite((sep,), rerun, what=str([files[i].name for i in group]))[0].cpu()
        for b, i in enumerate(group):
            for k in range(sep.shape[1]):
                d = args.output_dir / f"s{k}"
                d.mkdir(parents=True, exist_ok=True)
                wavio.save(d / f"{files[i].stem}.wav", sep[b, k:k + 1, :lens[b]], srs[b], bits=32)

    for j, group in enumerate(batches):
        w = j % K
        finish(w)  # the worker's previous batch
        items, srs = [], []
        for i in group:
            wav, sr = wavio.load(files[i])
            if sr != model_sr:  # the reference only warns (separate.py:151-155, quirk Q9)
                print(f"Warning: {files[i].stem}: this model expects {model_sr} Hz, but the file is {sr} Hz.")
            items.append((wav[:1], wav[:1]))
            srs.append(sr)
        mix, _, lens = datasets.pad_batch(items, side="right",
                                          to=eng.bucket_length(eng.padded_frames(max(lengths[i] for i in group))))
        sds = [seeds[i] for i in group] if seeds is not None else None
        with torch.cuda.stream(streams[w]):
            mix_d = mix.pin_memory().to(args.device, non_blocking=True)
            sep = separate_on_device(mix_d, models[w], kw, args.device, lengths=lens, seeds=sds)
        in_flight[w] = (group, lens, srs, sep, mix_d, sds)
    for w in range(K):
        finish(w)
    print(f"separated {len(files)} files into {args.output_dir}")


if __name__ == "__main__":
    main()
