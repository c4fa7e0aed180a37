#!/usr/bin/env python3
"""CPU probe (oracle): WHICH 16-bit tensors of the engine cost the agreement with fp32, and what does the storage format
buy?  One score evaluation of the oracle with the engine's rounding points emulated:
  w    weights of every convolution / NIN rounded
  a    the activated conv inputs (silu(GN(x)), what the staging writes to LDS) rounded
  t    the residual trunk rounded: block outputs (x + h) / sqrt 2, skip-stack entries, resampled x
  m    the tensors inside a block rounded (Conv_0 output h1)
  p    the output pyramid and its heads rounded
against the fp32 oracle, for bfloat16 (8 significand bits) and fp16 (11).  The engine's 16-bit mode is w+a+t+m+p.
Usage: python tools/probes/storage_dtype_probe.py [nf] [T]"""
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p_ in (os.path.join(ROOT, "diffusion-separation_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p_)
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
orig = dict(conv=F.conv2d, res=O._res_block, nin=O._nin, up=O.fir_up2, down=O.fir_down2)


def run(dt, what):
    q = lambda v: v.to(dt).float()
    qw = q if "w" in what else (lambda v: v)
    qa = q if "a" in what else (lambda v: v)
    qt = q if "t" in what else (lambda v: v)
    qm = q if "m" in what else (lambda v: v)
    qp = q if "p" in what else (lambda v: v)

    def res_block(pp, pre, x, temb, up=False, down=False):
        x = qt(x)  # the block input is a stored tensor of the trunk (or the in-place concat of two)
        h = qa(F.silu(O._gn(x, pp[pre + "GroupNorm_0.weight"], pp[pre + "GroupNorm_0.bias"])))
        if up:
            h, x = qa(orig["up"](h)), qt(orig["up"](x))
        elif down:
            h, x = qa(orig["down"](h)), qt(orig["down"](x))
        h = orig["conv"](h, qw(pp[pre + "Conv_0.weight"]), pp[pre + "Conv_0.bias"], padding=1)
        h = qm(h + F.linear(F.silu(temb), pp[pre + "Dense_0.weight"], pp[pre + "Dense_0.bias"])[:, :, None, None])
        h = qa(F.silu(O._gn(h, pp[pre + "GroupNorm_1.weight"], pp[pre + "GroupNorm_1.bias"])))
        h = orig["conv"](h, qw(pp[pre + "Conv_1.weight"]), pp[pre + "Conv_1.bias"], padding=1)
        if (pre + "Conv_2.weight") in pp:
            x = orig["conv"](x, qw(pp[pre + "Conv_2.weight"]), pp[pre + "Conv_2.bias"])
        return qt((x + h) / math.sqrt(2.0))

    def conv(x, w, b=None, *a, **k):  # the convolutions outside the blocks: stem, Combine, pyramid heads, output layer
        return qp(orig["conv"](x, qw(w), b, *a, **k))

    def nin(x, W, b):
        return qm(orig["nin"](qa(x), qw(W), b))

    O._res_block, O._nin, F.conv2d, O.F.conv2d = res_block, nin, conv, conv
    try:
        out = O.score_forward(p, cfg, xt, t, mixn)
    finally:
        O._res_block, O._nin, F.conv2d, O.F.conv2d = orig["res"], orig["nin"], orig["conv"], orig["conv"]
    return float(((out - ref).pow(2).mean() / ref.pow(2).mean()).sqrt())


print(f"one score evaluation, nf = {nf}, T = {T}: relative RMS against the fp32 oracle")
print(f"{'rounded tensors':42s} {'bf16':>10s} {'fp16':>10s}")
for name, what in (("everything (the 16-bit engine)", "watmp"), ("weights only", "w"), ("activated conv inputs only", "a"),
                   ("residual trunk only", "t"), ("in-block tensors only", "m"), ("pyramid / heads / stem only", "p"),
                   ("all but the residual trunk (fp32 trunk)", "wamp"), ("all but the weights", "atmp"),
                   ("all but the activated inputs", "wtmp")):
    print(f"{name:42s} {run(torch.bfloat16, what):10.3e} {run(torch.float16, what):10.3e}")
