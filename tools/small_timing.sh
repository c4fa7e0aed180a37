#!/bin/bash
# Build a profiling variant of the library with per-phase cycle counters in the small-image conv kernel
# (conv3x3_small.hip) and print the average cycles per block and phase.  (Run on the GPU box via gpurun.)
set -e
cd $(dirname $0)/../diffusion-separation_amd/csrc
mkdir -p ../abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -DSM_TIMING -c conv3x3_small.hip -o /tmp/small_timing.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/lib_timing.so /tmp/small_timing.o $(ls build/*.o | grep -Ev '/(conv3x3_small\.o)$')
cd ../..
DIFFSEP_LIB=$PWD/diffusion-separation_amd/abl/lib_timing.so python - <<'PY'
import ctypes, sys, os, torch
sys.path.insert(0, "diffusion-separation_amd")
from diffsep_amd import ops, _lib
l = ctypes.CDLL(os.environ["DIFFSEP_LIB"])
names = ["setup (geometry)", "issue loads of 2 phases", "GroupNorm table + fetch", "wait for the phase's loads", "activation", "barrier + LDS write + issue + barrier", "MFMA", "epilogue + store", "statistics"]
for (ci, co, H, W, act) in [(128, 128, 16, 16, 1), (128, 128, 16, 16, None), (128, 128, 8, 8, 1), (128, 128, 4, 4, 1), (128, 128, 4, 4, None)]:
    B = 16
    x = torch.randn(B, H, W, ci, device="cuda").to(torch.bfloat16)
    w = ops.pack_conv_weight(torch.randn(co, ci, 3, 3) / (9 * ci) ** 0.5, torch.bfloat16, chunk=32).cuda()
    b = torch.randn(co, device="cuda")
    sc = torch.rand(B, ci, device="cuda") + 0.5; sh = torch.randn(B, ci, device="cuda") * 0.1
    res = torch.randn(B, H, W, co, device="cuda").to(torch.bfloat16)
    st = torch.zeros(B, co, 2, dtype=torch.int64, device="cuda")
    y = torch.zeros(B, H, W, co, device="cuda", dtype=torch.bfloat16)
    kw = dict(gn=(sc, sh), gn_act=1) if act else {}
    run = lambda: ops.conv2d_fused(x, w, b, co, 3, res=res, out_scale=0.7071, w_chunk=32, out=y, stats=st, **kw)
    for _ in range(2): run()
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    l.diffsep_small_debug_read(out, 1)
    for _ in range(5): run()
    torch.cuda.synchronize()
    l.diffsep_small_debug_read(out, 1)
    nb = out[15]
    tot = sum(out[i] for i in range(11))
    print(f"{ci}->{co} {H}x{W} act={act}: {nb//5} blocks, {tot/nb:.0f} cycles/block in {out[11]/nb*10:.0f} ns = {tot/(out[11]*10):.2f} GHz shader clock")
    for i in range(9):
        print(f"    {names[i]:40s} {out[i]/nb:9.0f}  {100*out[i]/tot:5.1f} %")
PY
