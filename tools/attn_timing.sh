#!/bin/bash
# Per-phase cycle counters of the fused attention block (profiling build: -DATTN_TIMING), in the engine: one nf = 64 score
# evaluation at B = 16 (three 16 x 16 blocks and the 4 x 4 bottleneck block).  Run via gpurun.
set -e
cd $(dirname $0)/../diffusion-separation_amd/csrc
mkdir -p ../abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -DDS_HALF_F16 -DATTN_TIMING -c attn_fused.hip -o /tmp/attn_timing.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/lib_attntiming.so /tmp/attn_timing.o $(ls build_f16/*.o | grep -Ev '/(attn_fused\.o)$')
cd ../..
DIFFSEP_LIB_F16=$PWD/diffusion-separation_amd/abl/lib_attntiming.so python - <<'PY'
import ctypes, sys, os, torch
sys.path.insert(0, "diffusion-separation_amd")
from diffsep_amd import _lib, synth
from diffsep_amd.engine import Engine, pack_state_dict, param_table
l = ctypes.CDLL(os.environ["DIFFSEP_LIB_F16"])
names = ["issue: table operands, input tile, Wv + Wq fragments", "barrier 1 (table visible)", "affine + h -> LDS", "barrier 2", "V^T product -> LDS",
         "barrier 3", "Q and Q' products", "S = Q' h^T", "softmax", "P V (+ Wo fragments)", "output projection, residual, stores", "statistics"]
cfg = _lib.model_config(nf=64, num_sources=2, dtype=_lib.F16)
eng = Engine(cfg, pack_state_dict(cfg, synth.synth_state_dict([(n, s) for n, s, _ in param_table(cfg)], 7)))
eng.set_graph(False)
B, T = 16, 32000
x = torch.randn(B, 2, T, device="cuda") * 0.3; t = torch.full((B,), 0.5, device="cuda"); mix = torch.randn(B, 1, T, device="cuda")
for _ in range(2): eng.score(x, t, mix)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
l.diffsep_attn_debug_read(out, 1)
for _ in range(5): eng.score(x, t, mix)
torch.cuda.synchronize()
l.diffsep_attn_debug_read(out, 1)
nb = out[15]; tot = sum(out[i] for i in range(12))
print(f"{nb} blocks (5 evaluations x 4 attention blocks x {B} samples), {tot / nb:.0f} cycles per block (wave 0), average over the 16x16 and 4x4 blocks")
for i in range(12):
    print(f"    {names[i]:56s} {out[i] / nb:9.0f}  {100 * out[i] / tot:5.1f} %")
PY
