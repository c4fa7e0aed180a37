#!/usr/bin/env python3
"""Does running the B=16 sampler as two independent half batches on two streams (two engines, shared nothing but
the GPU) beat one B=16 launch sequence?  The low-resolution levels of the U-Net are ~100 tiny latency-bound launches
per NFE; a second stream can fill the machine meanwhile."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-separation_amd"))
from diffsep_amd import _lib, ops, synth
from diffsep_amd.engine import Engine, pack_state_dict, param_table
torch.set_grad_enabled(False)
B, T, S, N = 16, 32000, 2, 30
cfg = _lib.model_config(nf=64, num_sources=S, dtype=_lib.BF16)
blob = pack_state_dict(cfg, synth.synth_state_dict([(n, s) for n, s, _ in param_table(cfg)], 7))
sde = dict(ndim=S, d_lambda=2.0, sigma_min=0.05, sigma_max=0.5)
mix = torch.from_numpy(synth.synth_batch(B, T=T)[0]).cuda()
mix_norm, _, _ = ops.normalize_batch(mix)

def run(parts, full=False):
    engs = [Engine(cfg, blob) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    chunks = [mix_norm] * parts if full else list(mix_norm.chunk(parts, 0))
    def once(seed):
        outs = []
        for e, s, c in zip(engs, streams, chunks):
            with torch.cuda.stream(s):
                outs.append(e.pc_sample(c.contiguous(), sde, N=N, corrector_steps=1, snr=0.5, eps=0.03, denoise=True, seed=seed)[0])
        return outs
    once(1); once(1); torch.cuda.synchronize()   # eager pass, then graph capture
    t0 = time.perf_counter()
    for i in range(3): once(2 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    nb = sum(c.shape[0] for c in chunks)
    print(f"{parts} stream(s) x B={chunks[0].shape[0]}: {dt*1e3:.1f} ms per round -> {nb/dt:.2f} utt/s", flush=True)
    del engs

if len(sys.argv) > 1 and sys.argv[1] == "grid":   # K concurrent samplers of batch Bk each
    for Bk, Ks in ((48, (1,)), (32, (1, 2)), (16, (3, 4, 6)), (8, (4, 6, 8)), (4, (8,))):
        mix = torch.from_numpy(synth.synth_batch(Bk, T=T)[0]).cuda()
        mix_norm, _, _ = ops.normalize_batch(mix)
        for k in Ks:
            run(k, full=True)
    sys.exit(0)
for parts in [int(v) for v in (sys.argv[1].split(',') if len(sys.argv) > 1 else '1,2,4'.split(','))]:
    run(parts)
for parts in (2, 3):
    run(parts, full=True)
