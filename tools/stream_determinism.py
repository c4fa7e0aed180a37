"""Concurrency determinism probe of the score forward alone (eager launches): K streams x M calls on fixed inputs."""
import os, sys, time, torch
sys.path.insert(0, "diffusion-separation_amd")
from diffsep_amd import ops, synth, _lib
from diffsep_amd.engine import Engine, pack_state_dict, param_table
torch.set_grad_enabled(False)
K = int(os.environ.get("K", "4")); M = int(os.environ.get("M", "400")); B = int(os.environ.get("B", "1"))
cfg = _lib.model_config(nf=64, num_sources=2, dtype=_lib.BF16)
blob = pack_state_dict(cfg, synth.synth_state_dict([(n, s) for n, s, _ in param_table(cfg)], 7))
engs = [Engine(cfg, blob) for _ in range(K)]
streams = [torch.cuda.Stream() for _ in range(K)]
mns = [ops.normalize_batch(torch.from_numpy(synth.synth_batch(B, T=32000, start=i * B)[0]).cuda())[0] for i in range(K)]
g = torch.Generator().manual_seed(1)
xts = [(m.repeat(1, 2, 1) * 0.5 + 0.3 * torch.randn(B, 2, 32000, generator=g).cuda()) for m in mns]
ts = [torch.full((B,), 0.3 + 0.1 * w).cuda() for w in range(K)]
def one(w): return engs[w].score(xts[w], ts[w], mns[w])
ref = []
for w in range(K):
    one(w); ref.append(one(w)); torch.cuda.synchronize()
bad = [torch.zeros((), device="cuda", dtype=torch.int64) for _ in range(K)]
t0 = time.perf_counter()
for it in range(M):
    for w in range(K):
        with torch.cuda.stream(streams[w]):
            bad[w] += (one(w) != ref[w]).any()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"score-only K={K} M={M} B={B} env={ {k: v for k, v in os.environ.items() if k.startswith('DIFFSEP_')} }: "
      f"{sum(int(b) for b in bad)} / {K*M} mismatching, {K*M/dt:.0f} NFE/s", flush=True)
