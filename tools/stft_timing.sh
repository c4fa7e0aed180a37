#!/bin/bash
# Phase ticks (wave 0 of every block, s_memtime) of the fused STFT / iSTFT kernels: profiling build -DST_TIMING.  Run via gpurun.
set -e
cd $(dirname $0)/../diffusion-separation_amd/csrc
mkdir -p ../abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -DDS_HALF_F16 -DST_TIMING $ST_EXTRA -c stft.hip -o /tmp/st_timing.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/lib_sttiming.so /tmp/st_timing.o $(ls build_f16/*.o | grep -Ev '/(stft\.o)$')
cd ../..
DIFFSEP_LIB_F16=$PWD/diffusion-separation_amd/abl/lib_sttiming.so python - <<'PY'
import ctypes, os, sys, torch
sys.path.insert(0, "diffusion-separation_amd")
from diffsep_amd import ops
l = ctypes.CDLL(os.environ["DIFFSEP_LIB_F16"])
B, S, T, W = 16, 2, 32000, 256
x = torch.randn(B, S + 1, T, device="cuda") * 0.3
yy = (torch.randn(B, 256, W, 8, device="cuda") * 0.2).half()
out = (ctypes.c_ulonglong * 16)()
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); l.diffsep_st_debug_read(out, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    l.diffsep_st_debug_read(out, 1)
    return e0.elapsed_time(e1) / n * 1e3, [out[i] / n for i in range(8)]
us, t = timeit(lambda: ops.stft_pack(x[:, :S].contiguous(), x[:, S:].contiguous(), W, 8, shift=True, dtype=torch.float16))
nb = 2 * (W // 32) * B
print(f"stft_fused: {us:.1f} us per launch, {nb} blocks; ticks per block: staging {t[0]/nb:.0f}, products {t[1]/nb:.0f}, epilogue {t[2]/nb:.0f}")
us, t = timeit(lambda: ops.istft_unpack(yy, S, T))
nb = B * 9
print(f"istft_fused: {us:.1f} us per launch, {nb} blocks; ticks per block: U prologue {t[4]/nb:.0f}, products {t[5]/nb:.0f}, overlap-add stores {t[6]/nb:.0f}, envelope + output {t[7]/nb:.0f}")
PY
