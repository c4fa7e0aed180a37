#!/usr/bin/env python3
"""dtype="hybrid" with the f16 engine: agreement with the exact fp32 engine and time of one batch against the number of
reverse steps K that the split engine evaluates first (nf = 64 and 128, N = 30, B = 16 / 4, T = 32000).
usage: python tools/probes/hybrid_f16_probe.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "diffusion-separation_amd"))
from diffsep_amd import _lib, ops, synth  # noqa: E402
from diffsep_amd.engine import Engine, pack_state_dict, param_table  # noqa: E402

torch.set_grad_enabled(False)
sde = dict(ndim=2, d_lambda=2.0, sigma_min=0.05, sigma_max=0.5)


def si_sdr(est, ref):
    est, ref = est.double(), ref.double()
    a = (est * ref).sum(-1, keepdim=True) / (ref * ref).sum(-1, keepdim=True)
    return 10 * torch.log10(((a * ref) ** 2).sum(-1) / ((est - a * ref) ** 2).sum(-1))


for nf, B in ((64, 16), (128, 4)):
    mk = lambda dt, kind=None: Engine(_lib.model_config(nf=nf, num_sources=2, dtype=dt, spec_factor=0.33 if nf == 64 else 0.15), blob, lib_kind=kind)
    cfg = _lib.model_config(nf=nf, num_sources=2, dtype=_lib.F32, spec_factor=0.33 if nf == 64 else 0.15)
    blob = pack_state_dict(cfg, synth.synth_state_dict([(n, s) for n, s, _ in param_table(cfg)], 7))
    e32, e16, esp = mk(_lib.F32), mk(_lib.F16), mk(_lib.F32_SPLIT, "f16")
    mix = torch.from_numpy(synth.synth_batch(B, T=32000)[0]).cuda()
    mn, _, _ = ops.normalize_batch(mix)
    kw = dict(N=30, corrector_steps=1, snr=0.5, eps=0.03, denoise=True, seed=11)
    ref, _ = e32.pc_sample(mn, sde, **kw)
    for K in (0, 2, 3, 5, 7, 10):
        run = (lambda: e16.pc_sample(mn, sde, tail=esp, head_steps=K, **kw)[0]) if K else (lambda: e16.pc_sample(mn, sde, **kw)[0])
        run(); run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = run()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        s = si_sdr(out, ref)
        print(f"nf {nf} B {B} head steps {K:2d}: SI-SDR vs fp32 mean {float(s.mean()):6.2f} min {float(s.min()):6.2f} dB, one batch alone {dt * 1e3:7.1f} ms = {B / dt:6.2f} utt/s")
    del e32, e16, esp
