"""Locate the first tensor that differs under stream concurrency: per-allocation comparison of the engine's workspace
arena (Engine.debug_arena) against a single-stream reference.
  1. DIFFSEP_DBG_ALLOC=1 K=1 M=0 python tools/stream_first_diff.py 2>&1 | grep "diffsep alloc" > alloc.txt
  2. ALLOC=alloc.txt M=100 python tools/stream_first_diff.py
(the [diffsep stats] lines of step 1 map accumulator offsets to the convolutions that fill them)"""
import os, sys, time, torch
sys.path.insert(0, "diffusion-separation_amd")
from diffsep_amd import ops, synth, _lib
from diffsep_amd.engine import Engine, pack_state_dict, param_table
torch.set_grad_enabled(False)
K = int(os.environ.get("K", "4")); M = int(os.environ.get("M", "300"))
cfg = _lib.model_config(nf=64, num_sources=2, dtype=_lib.BF16)
blob = pack_state_dict(cfg, synth.synth_state_dict([(n, s) for n, s, _ in param_table(cfg)], 7))
engs = [Engine(cfg, blob) for _ in range(K)]
streams = [torch.cuda.Stream() for _ in range(K)]
mns = [ops.normalize_batch(torch.from_numpy(synth.synth_mixture(i, T=32000, fs=8000, n_src=2)[0])[None].cuda())[0] for i in range(K)]
g = torch.Generator().manual_seed(1)
xts = [(m.repeat(1, 2, 1) * 0.5 + 0.3 * torch.randn(1, 2, 32000, generator=g).cuda()) for m in mns]
ts = [torch.full((1,), 0.3 + 0.1 * w).cuda() for w in range(K)]
def one(w): return engs[w].score(xts[w], ts[w], mns[w])
views, refs = [], []
for w in range(K):
    one(w); one(w); torch.cuda.synchronize()
    v, fb = engs[w].debug_arena()
    v = v[fb:].view(torch.int64)
    views.append(v); refs.append(v.clone())
    one(w); torch.cuda.synchronize()
    assert torch.equal(v, refs[w]), "single-stream forward is not reproducible"
print("arena forward region:", views[0].numel() * 8 / 1e6, "MB; fwd_base", fb)
allocs = [tuple(int(v) for v in l.split()[2:4]) for l in open(os.environ["ALLOC"])]
i0 = [i for i, (o, b) in enumerate(allocs) if o == fb][0]
i1 = [i for i, (o, b) in enumerate(allocs) if o == fb][1] if sum(o == fb for o, b in allocs) > 1 else len(allocs)
allocs = allocs[i0:i1]
nw = views[0].numel()
seg = torch.zeros(nw, dtype=torch.int64, device="cuda")
for j, (o, b) in enumerate(allocs):
    seg[(o - fb) // 8:(o - fb + b + 7) // 8] = j
cnt = [torch.zeros((M, len(allocs)), device="cuda") for _ in range(K)]
for it in range(M):
    for w in range(K):
        with torch.cuda.stream(streams[w]):
            one(w)
            neq = (views[w] != refs[w]).float()
            cnt[w][it].index_add_(0, seg, neq)
torch.cuda.synchronize()
for w in range(K):
    c = cnt[w].cpu()
    for it in range(M):
        if c[it].sum() > 0:
            nz = [(j, allocs[j][0], allocs[j][1], int(c[it, j])) for j in range(len(allocs)) if c[it, j] > 0]
            print(f"stream {w} it {it}: {len(nz)} allocations differ; first: {nz[:6]}")
