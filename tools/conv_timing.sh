#!/bin/bash
# Build a profiling variant of the library with per-phase cycle counters in the conv kernel and print the
# average cycles per block and phase for a few layer shapes.  (Run on the GPU box via gpurun.)
set -e
cd $(dirname $0)/../diffusion-separation_amd/csrc
mkdir -p ../abl
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCONV_TIMING -c conv_mfma.hip -o /tmp/conv_timing.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/lib_timing.so /tmp/conv_timing.o build/norm.o build/stft.o build/sde.o build/engine.o
cd ../..
DIFFSEP_LIB=$PWD/diffusion-separation_amd/abl/lib_timing.so python - <<'PY'
import ctypes, sys, os, torch
sys.path.insert(0, "diffusion-separation_amd")
from diffsep_amd import ops, _lib
l = ctypes.CDLL(os.environ["DIFFSEP_LIB"])
names = ["issue first loads", "wait+LDS store+2 barriers", "issue next loads", "MFMA loop", "acc dump+barriers+residual loads", "epilogue LDS read+math", "epilogue global stores"]
for (k, ci, co, H, W) in [(3, 64, 64, 256, 256), (3, 128, 64, 256, 256), (3, 128, 128, 64, 64), (1, 128, 64, 256, 256)]:
    B = 16
    x = torch.randn(B, H, W, ci, device="cuda").to(torch.bfloat16)
    w = (torch.randn(co, k * k, ci, device="cuda") / (k * k * ci) ** 0.5).to(torch.bfloat16)
    b = torch.randn(co, device="cuda")
    for _ in range(2): ops.conv2d(x, w, b, co, k)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 8)()
    l.diffsep_debug_read(out, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.conv2d(x, w, b, co, k)
    e1.record(); torch.cuda.synchronize()
    l.diffsep_debug_read(out, 1)
    nb = out[7]
    tot = sum(out[i] for i in range(7))
    print(f"k{k} {ci}->{co} {H}x{W}: {e0.elapsed_time(e1)/5*1e3:.1f} us/launch, {nb//5} blocks, {tot/nb:.0f} cycles/block (clock64 = 100 MHz ticks?)")
    for i in range(7):
        print(f"    {names[i]:28s} {out[i]/nb:9.0f}  {100*out[i]/tot:5.1f} %")
PY
