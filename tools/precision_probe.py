#!/usr/bin/env python3
"""Agreement of the f16 engine (library variant in DIFFSEP_LIB_F16) with the exact fp32 engine after a full sampler run, the
figures of bench.py's `precision` object, for A/B runs of kernel variants:  python tools/precision_probe.py [nf] [B] [spec_factor]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-separation_amd"))
from diffsep_amd import _lib, ops, synth
from diffsep_amd.engine import Engine, pack_state_dict, param_table
torch.set_grad_enabled(False)
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sf = float(sys.argv[3]) if len(sys.argv) > 3 else (0.33 if nf <= 64 else 0.15)
T, S = 32000, 2
cfg = lambda dt: _lib.model_config(nf=nf, num_sources=S, dtype=dt, spec_factor=sf)
sd = synth.synth_state_dict([(n, s) for n, s, _ in param_table(cfg(_lib.F16))], 7)
blob = pack_state_dict(cfg(_lib.F16), sd)
mix = torch.from_numpy(synth.synth_batch(B, T=T)[0]).cuda()
mn = ops.normalize_batch(mix)[0]
sde = dict(ndim=S, d_lambda=2.0, sigma_min=0.05, sigma_max=0.5)
kw = dict(N=30, corrector_steps=1, snr=0.5, eps=0.03, denoise=True, seed=4242)
def si_sdr_db(est, ref):
    est, ref = est.double(), ref.double()
    a = (est * ref).sum(-1, keepdim=True) / (ref * ref).sum(-1, keepdim=True)
    return 10 * torch.log10(((a * ref) ** 2).sum(-1) / ((est - a * ref) ** 2).sum(-1))
e32 = Engine(cfg(_lib.F32), blob)
o32 = ops.scale_output(mix, e32.pc_sample(mn, sde, **kw)[0])
e32.close()
e16 = Engine(cfg(_lib.F16), blob)
o16 = ops.scale_output(mix, e16.pc_sample(mn, sde, **kw)[0])
q = si_sdr_db(o16, o32)
rel = float(((o16 - o32).double().pow(2).mean() / o32.double().pow(2).mean()).sqrt())
print(f"nf={nf} B={B} lib={os.path.basename(os.environ.get('DIFFSEP_LIB_F16', 'shipped'))}: f16 vs fp32 engine SI-SDR mean {float(q.mean()):.2f} min {float(q.min()):.2f} dB, "
      f"rel rms {rel:.3e}, abs rms {float((o16 - o32).double().pow(2).mean().sqrt()):.3e}; per utterance min {[round(float(v), 1) for v in q.min(-1).values]}")
