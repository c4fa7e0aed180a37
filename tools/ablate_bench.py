#!/usr/bin/env python3
"""What would a step cost if a class of launches were free?  The bench.py step (B = 16 x 4 s, nf = 64, f16, N = 30 + 1 corrector
step) with classes of launches SKIPPED through the engine option "ablate" (engine.hip `ablated`): the results are garbage,
the buffers keep the realistic values of the un-ablated warm-up, and the drop in time is the upper bound of what any
optimisation of that class can buy — for one batch alone and with K batches in flight.

    python tools/ablate_bench.py [--in-flight 4] [--steps 6] [--masks 0,1,2,4,...]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-separation_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

NAMES = {1: "convs/GEMMs <= 16 rows (attention incl.)", 2: "convs 32 rows", 4: "convs 64 rows", 8: "gn_apply / FIR resampling",
         16: "gn_finalize", 32: "convs 128 rows", 64: "convs 256 rows", 128: "STFT / iSTFT"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--in-flight", type=int, default=4)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--nf", type=int, default=64)
    ap.add_argument("--masks", default="0,1,2,4,8,16,32,64,128,7,31,0")
    ap.add_argument("--options", default="", help="engine options for every run, e.g. kc64=1,no_rw128=1")
    args = ap.parse_args()
    from diffsep_amd import _lib, ops, synth
    from diffsep_amd.engine import Engine, pack_state_dict, param_table
    torch.set_grad_enabled(False)
    B, T, S, K = 16, 32000, 2, args.in_flight
    cfg = _lib.model_config(nf=args.nf, num_sources=S, dtype=_lib.F16)
    blob = pack_state_dict(cfg, synth.synth_state_dict([(n, s) for n, s, _ in param_table(cfg)], 7))
    engs = [Engine(cfg, blob) for _ in range(K)]
    for kv in [v for v in args.options.split(",") if v]:
        for e in engs:
            e.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    streams = [torch.cuda.Stream() for _ in range(K)]
    mix = torch.from_numpy(synth.synth_batch(B, T=T)[0]).cuda()
    sde = dict(ndim=S, d_lambda=2.0, sigma_min=0.05, sigma_max=0.5)
    keep = [None] * K

    def step(i, w):
        with torch.cuda.stream(streams[w]):
            mn, _, _ = ops.normalize_batch(mix)
            sep, _ = engs[w].pc_sample(mn, sde, N=30, corrector_steps=1, snr=0.5, eps=0.03, denoise=True, seed=1000 + i)
            keep[w] = (mn, sep, ops.scale_output(mix, sep))

    rows = []
    for mask in [int(m) for m in args.masks.split(",")]:
        for e in engs:
            e.set_option("ablate", mask)
        for w in range(K):  # plan (first) / graph capture
            step(w, w)
            step(w, w)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(0, 0)
        torch.cuda.synchronize()
        alone = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i, i % K)
        torch.cuda.synchronize()
        per = (time.perf_counter() - t0) / args.steps * 1e3
        rows.append({"mask": mask, "skipped": [NAMES[b] for b in NAMES if mask & b], "one_batch_alone_ms": round(alone, 1),
                     "ms_per_step_in_flight": round(per, 1), "utt_per_s": round(B / per * 1e3, 2)})
        print(json.dumps(rows[-1]), flush=True)
    base = rows[0]
    print("\n| skipped | one batch alone ms (saved) | ms per step, %d in flight (saved) | utt/s |\n|---|---:|---:|---:|" % K)
    for r in rows:
        print("| %s | %.1f (%.1f) | %.1f (%.1f) | %.1f |" % (" + ".join(r["skipped"]) or "nothing", r["one_batch_alone_ms"],
              base["one_batch_alone_ms"] - r["one_batch_alone_ms"], r["ms_per_step_in_flight"],
              base["ms_per_step_in_flight"] - r["ms_per_step_in_flight"], r["utt_per_s"]))


if __name__ == "__main__":
    main()
