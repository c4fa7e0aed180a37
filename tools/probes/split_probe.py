"""DIFFSEP_F32_SPLIT (fp32 tensors, every MFMA product as 3 bf16 MFMAs on hi / lo halves) against the exact fp32 engine:
agreement of one score evaluation and of the whole sampler, speed, and its use as the fp32 head of the hybrid schedule.
Run on the GPU box: python tools/probes/split_probe.py"""
import os, sys, time
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "diffusion-separation_amd"))
from diffsep_amd import _lib, ops, synth
from diffsep_amd.engine import Engine, pack_state_dict, param_table

DEV = torch.device("cuda")
SDE = dict(ndim=2, d_lambda=2.0, sigma_min=0.05, sigma_max=0.5)


def si_sdr(est, ref):
    a = (est * ref).sum(-1, keepdim=True) / (ref * ref).sum(-1, keepdim=True)
    return 10 * torch.log10((a * ref).pow(2).sum(-1) / (est - a * ref).pow(2).sum(-1))


def rel(a, b):
    return float(((a - b).pow(2).mean() / b.pow(2).mean()).sqrt())


def main():
    B, T, N = 16, 32000, 30
    engs = {}
    for name, dt in (("f32", _lib.F32), ("split", _lib.F32_SPLIT), ("bf16", _lib.BF16)):
        cfg = _lib.model_config(nf=64, num_sources=2, dtype=dt)
        sd = synth.synth_state_dict([(n, s) for n, s, _ in param_table(cfg)], 1)
        engs[name] = Engine(cfg, pack_state_dict(cfg, sd))
    mix = torch.from_numpy(synth.synth_batch(B, T=T)[0]).to(DEV)
    mixn = ops.normalize_batch(mix)[0]
    g = torch.Generator(device=DEV).manual_seed(3)
    xt = torch.randn(B, 2, T, device=DEV, generator=g) * 0.5 + mixn / 2
    t = torch.full((B,), 0.5, device=DEV)
    s = {k: e.score(xt, t, mixn) for k, e in engs.items()}
    print(f"one score evaluation, rel. RMS against exact fp32: split {rel(s['split'], s['f32']):.3e}   bf16 {rel(s['bf16'], s['f32']):.3e}")
    outs, times = {}, {}
    for k, e in engs.items():
        e.pc_sample(mixn, SDE, N=2, corrector_steps=1, seed=7)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs[k], _ = e.pc_sample(mixn, SDE, N=N, corrector_steps=1, seed=7)
        torch.cuda.synchronize()
        times[k] = time.perf_counter() - t0
    for k in ("split", "bf16"):
        sd_ = si_sdr(outs[k], outs["f32"])
        print(f"sampler N={N}: {k:5s} vs fp32: rel RMS {rel(outs[k], outs['f32']):.3e}, SI-SDR mean {float(sd_.mean()):.1f} min {float(sd_.min()):.1f} dB")
    print("one batch of 16 alone: " + ", ".join(f"{k} {times[k]*1e3:.0f} ms = {B/times[k]:.1f} utt/s" for k in engs))
    for head, K in (("f32", 10), ("split", 10), ("split", 5), ("split", 6), ("split", 7), ("split", 8), ("split", 9), ("split", 15), ("split", 30)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        o, _ = engs["bf16"].pc_sample(mixn, SDE, N=N, corrector_steps=1, seed=7, tail=engs[head], head_steps=K)
        torch.cuda.synchronize()
        dtm = time.perf_counter() - t0
        sd_ = si_sdr(o, outs["f32"])
        print(f"hybrid, first {K} steps on the {head} engine: SI-SDR vs fp32 mean {float(sd_.mean()):.1f} min {float(sd_.min()):.1f} dB, {B/dtm:.1f} utt/s (one batch alone)")


if __name__ == "__main__":
    main()
