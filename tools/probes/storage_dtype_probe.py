#!/usr/bin/env python3
"""CPU probe (oracle): how much of the 16-bit engine's error comes from the STORAGE format?  One score evaluation of the
oracle with every convolution's input, weights and output rounded to bf16 / fp16 (what the engine's 16-bit tensors
do), against the fp32 oracle.  fp16 keeps 11 significand bits against bf16's 8.
Usage: python tools/probes/storage_dtype_probe.py [nf] [T]"""
import os
import sys
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "diffusion-separation_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import diffsep_oracle as O  # noqa: E402
from diffsep_amd import synth  # noqa: E402

torch.set_grad_enabled(False)
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16000
cfg = O.default_config(nf, 2)
p = O.to_torch(synth.synth_state_dict(O.param_table(cfg), 7))
mix = torch.from_numpy(synth.synth_batch(1, T=T)[0])
mixn, _, _ = O.normalize_batch(mix)
xt = O.prior_sampling(cfg, mixn, torch.from_numpy(synth.synth_noise("probe.z", (1, 2, T))))
t = torch.tensor([0.7])
ref = O.score_forward(p, cfg, xt, t, mixn)
conv0 = F.conv2d


def run(dt):
    q = lambda v: v.to(dt).float()

    def conv(x, w, b=None, *a, **k):
        return q(conv0(q(x), q(w), b, *a, **k))
    F.conv2d = conv
    O.F.conv2d = conv
    try:
        out = O.score_forward(p, cfg, xt, t, mixn)
    finally:
        F.conv2d = conv0
        O.F.conv2d = conv0
    rel = float(((out - ref).pow(2).mean() / ref.pow(2).mean()).sqrt())
    return rel


for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
    print(f"{name}: relative RMS of one score evaluation vs fp32 = {run(dt):.3e}")
