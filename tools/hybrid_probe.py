#!/usr/bin/env python3
"""Where along the reverse trajectory does bf16 cost quality?  SI-SDR of the separated waveforms against the fp32
engine's output (same seeds) when the fp32 engine evaluates the first H and / or the last K reverse steps.
Usage: python tools/hybrid_probe.py [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-separation_amd"))
from diffsep_amd import _lib, ops, synth  # noqa: E402
from diffsep_amd.engine import Engine, pack_state_dict, param_table  # noqa: E402

torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
T, N, S = 32000, 30, 2
sde = dict(ndim=S, d_lambda=2.0, sigma_min=0.05, sigma_max=0.5)


def si_sdr(est, ref):
    est, ref = est.double(), ref.double()
    a = (est * ref).sum(-1, keepdim=True) / (ref * ref).sum(-1, keepdim=True)
    return 10 * torch.log10(((a * ref) ** 2).sum(-1) / ((est - a * ref) ** 2).sum(-1))


engs = {}
for name, dt in (("bf16", _lib.BF16), ("f32", _lib.F32)):
    cfg = _lib.model_config(nf=64, num_sources=S, dtype=dt)
    sd = synth.synth_state_dict([(n, s) for n, s, _ in param_table(cfg)], 7)
    engs[name] = Engine(cfg, pack_state_dict(cfg, sd))
mix = torch.from_numpy(synth.synth_batch(B, T=T)[0]).cuda()
mixn = ops.normalize_batch(mix)[0]
kw = dict(N=N, corrector_steps=1, snr=0.5, eps=0.03, denoise=True, seed=4242)
ref = engs["f32"].pc_sample(mixn, sde, **kw)[0]
print("ref finite", bool(torch.isfinite(ref).all()), "rms", float(ref.pow(2).mean().sqrt()))
for head, tail in ((0, 0), (0, 5), (0, 15), (0, 25), (0, 29), (5, 0), (10, 0), (15, 0), (20, 0), (25, 0), (29, 0), (1, 0), (2, 0), (0, 30)):
    out = engs["bf16"].pc_sample(mixn, sde, tail=engs["f32"], tail_steps=tail, head_steps=head, **kw)[0]
    q = si_sdr(out, ref)
    print(f"fp32 head {head:2d} tail {tail:2d}: finite {bool(torch.isfinite(out).all())}  SI-SDR mean {float(q.mean()):6.2f} min {float(q.min()):6.2f} dB")
# per-step error growth: run bf16 for the first k steps only, fp32 after
