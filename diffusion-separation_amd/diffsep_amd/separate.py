"""separate.py — same command line as the reference's separate.py:102-162:

    python -m diffsep_amd.separate input_dir output_dir [--model CKPT] [-d cuda:0] [-N 30] [--snr 0.5]
           [--corrector-steps 1] [--denoise True] [-s SCHEDULE]

For every *.wav in input_dir: load -> normalize_batch -> reverse-diffusion PC sampler on the HIP engine ->
scale_output -> write output_dir/s{i}/name.wav (directories s0, s1 like separate.py:157-158).
Additions: --synthetic-weights NF runs with random-init weights of width NF when no checkpoint is
available (there is no network here: the HF default 'fakufaku/diffsep' cannot be downloaded), --dtype,
--batch to separate several equal-length files per engine call, --streams K to keep K files (or batches) in flight on
K engines / HIP streams (one file at a time leaves most of the GPU idle: 6.0 -> 18.5 files/s at K = 4 for 4 s files)
and --seed (with it the outputs do not depend on K).
"""
import argparse
import os
from pathlib import Path

import torch

from . import ops, wavio
from .pl_model import DiffSepModel, cfg_get, default_config

DEFAULT_MODEL = "fakufaku/diffsep"


def get_model(args):
    if args.synthetic_weights:
        model = DiffSepModel(default_config(nf=args.synthetic_weights), dtype=args.dtype, device=args.device)
    else:
        path = Path(args.model)
        if not path.exists():
            raise FileNotFoundError(f"checkpoint '{args.model}' not found (Hugging Face download needs network access; "
                                    "pass a local Lightning checkpoint or --synthetic-weights NF)")
        model = DiffSepModel.load_from_checkpoint(str(path), dtype=args.dtype, device=args.device)
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


def separate_on_device(mix, model, sampler_kwargs, device):
    """Enqueue the separation of mix [1,T] / [B,1,T] on the current stream; returns the device tensor [B,S,T]."""
    mix = mix.to(device)
    if mix.dim() == 2:
        mix = mix[None]
    (mix_norm, _), *_ = model.normalize_batch((mix, None))
    sampler = model.get_pc_sampler("reverse_diffusion", "ald2", mix_norm, **sampler_kwargs)
    with torch.no_grad():
        sep, nfe, *_ = sampler()
    return scale_output(mix, sep)


def separate(mix, model, sampler_kwargs, device):
    """mix [1,T] (one file, like the reference) or [B,1,T] (a batch of equal-length files)."""
    return separate_on_device(mix, model, sampler_kwargs, device).cpu()


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
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--streams", type=int, default=1, help="files (batches) in flight: K engines on K HIP streams")
    ap.add_argument("--seed", type=int, default=None, help="torch.manual_seed before the first file")
    args = ap.parse_args(argv)
    K = max(1, args.streams)
    if K > 1:  # (see evaluate.py: hardware queues; must precede the first torch.cuda call)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if not torch.cuda.is_available():
        raise SystemExit("No GPU visible: this build has no CPU path (the reference falls back to CPU here)")
    torch.cuda.set_device(torch.device(args.device))
    model, kw = get_model(args)
    models = [model] + [get_model(args)[0] for _ in range(K - 1)]
    for m in models:
        m.score_model.engine()  # engines before streams (hardware queues are handed out in creation order)
    if args.seed is not None:
        torch.manual_seed(args.seed)
    model_sr = cfg_get(model.config, "model.fs", 8000)
    if args.output_dir.is_file():
        raise ValueError("Output directory is a file")
    args.output_dir.mkdir(parents=True, exist_ok=True)
    files = sorted(args.input_dir.glob("*.wav"))
    if files:  # workspace for the longest file now (growing it later synchronises the whole device)
        longest = max(wavio.info(p)[1] for p in files)
        for m in models:
            m.score_model.engine().reserve(args.batch, longest)
    streams = [torch.cuda.Stream() for _ in range(K)] if K > 1 else [torch.cuda.current_stream()]
    pending = []
    in_flight = [None] * K  # per worker: (group, device result) of the batch running on its stream
    n_groups = 0

    def finish(w):
        if in_flight[w] is None:
            return
        group, sep = in_flight[w]
        in_flight[w] = None
        streams[w].synchronize()
        for (p, _, sr), s in zip(group, sep.cpu()):
            for i in range(s.shape[0]):
                d = args.output_dir / f"s{i}"
                d.mkdir(parents=True, exist_ok=True)
                wavio.save(d / f"{p.stem}.wav", s[i:i + 1], sr)

    def flush():
        nonlocal n_groups
        if not pending:
            return
        w = n_groups % K
        n_groups += 1
        finish(w)  # the worker's previous batch
        mix = torch.stack([wv for _, wv, _ in pending])  # [B,1,T]
        with torch.cuda.stream(streams[w]):
            in_flight[w] = (list(pending), separate_on_device(mix, models[w], kw, args.device))
        pending.clear()

    for p in files:
        wav, sr = wavio.load(p)
        if sr != model_sr:  # the reference only warns (separate.py:151-155, quirk Q9)
            print(f"Warning: {p.stem}: this model expects {model_sr} Hz, but the file is {sr} Hz.")
        wav = wav[:1]
        if pending and (pending[0][1].shape[-1] != wav.shape[-1] or len(pending) >= args.batch):
            flush()
        pending.append((p, wav, sr))
        if len(pending) >= args.batch:
            flush()
    flush()
    for w in range(K):
        finish(w)
    print(f"separated {len(files)} files into {args.output_dir}")


if __name__ == "__main__":
    main()
