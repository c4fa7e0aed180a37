#!/bin/bash
# A/B of the fused inverse transform (istft_fused_kernel): prefetch depth of the DFT fragment stream and sources per block.
# Run via gpurun from the repo root; prints microseconds per launch (B = 16, S = 2, T = 32000) for every variant.
set -e
cd ${GRAFT_REPO_ROOT:-.}/diffusion-separation_amd/csrc
mkdir -p ../abl
for V in "base:" "d4:-DSI_DEPTH=4" "d5:-DSI_DEPTH=5" "d6:-DSI_DEPTH=6" "ns1:-DSI_NS1" "ns1d5:-DSI_NS1 -DSI_DEPTH=5"; do
  NAME=${V%%:*}; FL=${V#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -DDS_HALF_F16 $FL -c stft.hip -o /tmp/st_$NAME.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/lib_st_$NAME.so /tmp/st_$NAME.o $(ls build_f16/*.o | grep -Ev '/(stft\.o)$')
done
cd ../..
for NAME in base d4 d5 d6 ns1 ns1d5 base; do
DIFFSEP_LIB_F16=$PWD/diffusion-separation_amd/abl/lib_st_$NAME.so python - $NAME <<'PY'
import sys, torch
sys.path.insert(0, "diffusion-separation_amd")
from diffsep_amd import ops
B, S, T, W = 16, 2, 32000, 256
yy = (torch.randn(B, 256, W, 8, device="cuda") * 0.2).half()
ref = None
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
us = timeit(lambda: ops.istft_unpack(yy, S, T))
out = ops.istft_unpack(yy, S, T)
print(f"istft {sys.argv[1]:8s} {us:7.1f} us   checksum {float(out.double().sum()):.6f} {float(out.double().abs().sum()):.6f}")
PY
done
