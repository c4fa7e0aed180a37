#!/usr/bin/env python3
"""Every MFMA launch of one batch (60 evaluations, B = 16, T = 32000) by (kernel instantiation, shape): launches, average
duration (eager launches with events: diffsep_engine_profile_records), fractions of the MFMA and HBM roofs.
usage: python tools/shape_table.py [nf] [dtype f16|bf16|f32|split] [B] [N]   -> markdown on stdout"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-separation_amd"))
from diffsep_amd import _lib, ops, synth  # noqa: E402
from diffsep_amd.engine import Engine, pack_state_dict, param_table  # noqa: E402

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dt = sys.argv[2] if len(sys.argv) > 2 else "f16"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
N = int(sys.argv[4]) if len(sys.argv) > 4 else 30
cfg = _lib.model_config(nf=nf, num_sources=2, dtype={"f16": _lib.F16, "bf16": _lib.BF16, "f32": _lib.F32, "split": _lib.F32_SPLIT}[dt], spec_factor=0.33 if nf == 64 else 0.15)
sd = synth.synth_state_dict([(n, s) for n, s, _ in param_table(cfg)], 7)
eng = Engine(cfg, pack_state_dict(cfg, sd))
mix = torch.from_numpy(synth.synth_batch(B, T=32000)[0]).cuda()
mn, _, _ = ops.normalize_batch(mix)
sde = dict(ndim=2, d_lambda=2.0, sigma_min=0.05, sigma_max=0.5)
eng.set_graph(False)
eng.pc_sample(mn, sde, N=2, seed=1)
eng.profile_begin()
eng.pc_sample(mn, sde, N=N, seed=1)
eng.profile_end()
agg = {}
for r in eng.profile_records():
    key = (r["kernel"], r["taps"], r["Cin"], r["Cout"], r["H"], r["W"], r["skip_cin"], r["has_res"])
    a = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
    a[0] += 1; a[1] += r["ms"]; a[2] += r["flops"]; a[3] += r["bytes"]
tot = sum(a[1] for a in agg.values())
PEAK = {"f16": 2.5e15, "bf16": 2.5e15, "f32": 157.3e12, "split": 2.5e15 / 3}[dt]  # dense MFMA peak of the mode (split: 3 bf16 MFMAs per product)
nfe = 2 * N
print(f"# MFMA launches of one batch: nf = {nf}, {dt}, B = {B}, {nfe} evaluations, eager launches timed with events; total {tot:.1f} ms\n")
print("| kernel | shape | launches / evaluation | avg us | ms / batch | % | of MFMA peak | of HBM peak |")
print("|---|---|---:|---:|---:|---:|---:|---:|")
for (kn, taps, ci, co, H, W, sk, hr), (n, ms, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    k = 3 if taps == 9 else 1
    shape = f"{k}x{k} {ci}->{co}" + (f" +skip{sk}" if sk else "") + (" +res" if hr else "") + f" @{H}x{W}"
    print(f"| `{kn}` | {shape} | {n / nfe:.2f} | {ms / n * 1e3:.1f} | {ms:.2f} | {100 * ms / tot:.1f} | {fl / (ms * 1e-3) / PEAK:.3f} | {by / (ms * 1e-3) / 8e12:.3f} |")
