#!/bin/bash
# A/B of the fused transforms (stft_fused_kernel / istft_fused_kernel): waves per block, prefetch depth of the DFT fragment stream,
# sources per block.  Run via gpurun from the repo root; prints microseconds per launch (B = 16, S = 2, T = 32000) and a checksum
# of the output for every variant (the variants compute the same bits).
#   VARIANTS="base: nw16:-DSI_NW=16" bash tools/istft_ab.sh
set -e
cd ${GRAFT_REPO_ROOT:-.}/diffusion-separation_amd/csrc
mkdir -p ../abl
VARIANTS=${VARIANTS:-"ship: si4:-DSI_NW=4 si8:-DSI_NW=8 sf4:-DSF_NW=4 sf8:-DSF_NW=8 ship:"}
for V in $VARIANTS; do
  NAME=${V%%:*}; FL=$(echo ${V#*:} | tr ',' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -DDS_HALF_F16 $FL -c stft.hip -o /tmp/st_$NAME.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/lib_st_$NAME.so /tmp/st_$NAME.o $(ls build_f16/*.o | grep -Ev '/(stft\.o)$')
done
cd ../..
for V in $VARIANTS; do
NAME=${V%%:*}
DIFFSEP_LIB_F16=$PWD/diffusion-separation_amd/abl/lib_st_$NAME.so python - $NAME <<'PY'
import sys, torch
sys.path.insert(0, "diffusion-separation_amd")
from diffsep_amd import ops
B, S, T, W = 16, 2, 32000, 256
torch.manual_seed(0)
yy = (torch.randn(B, 256, W, 8, device="cuda") * 0.2).half()
x = torch.randn(B, S + 1, T, device="cuda") * 0.3
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
inv = lambda: ops.istft_unpack(yy, S, T)
fwd = lambda: ops.stft_pack(x[:, :S].contiguous(), x[:, S:].contiguous(), W, 8, shift=True, dtype=torch.float16)
ui, uf = timeit(inv), timeit(fwd)
oi, of = inv(), fwd()
print(f"{sys.argv[1]:10s} istft {ui:6.1f} us (sum {float(oi.double().sum()):.6f} abs {float(oi.double().abs().sum()):.4f})   "
      f"stft {uf:6.1f} us (sum {float(of.double().sum()):.4f} abs {float(of.double().abs().sum()):.3f})")
PY
done
