#!/bin/bash
# Per-phase cycle counters of the weight-stationary conv kernel (profiling build).  Run via gpurun.
set -e
cd $(dirname $0)/../diffusion-separation_amd/csrc
mkdir -p ../abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -DWS_TIMING -c conv3x3_ws.hip -o /tmp/ws_timing.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/lib_wstiming.so /tmp/ws_timing.o $(ls build/*.o | grep -Ev '/(conv3x3_ws\.o)$')
cd ../..
DIFFSEP_LIB=$PWD/diffusion-separation_amd/abl/lib_wstiming.so python - <<'PY'
import ctypes, sys, os, torch
sys.path.insert(0, "diffusion-separation_amd")
from diffsep_amd import ops
l = ctypes.CDLL(os.environ["DIFFSEP_LIB"])
names = ["barrier wait", "issue loads", "MFMA + activation", "LDS write", "prologue", "tail (stats)", "epilogue", "-", "-", "-", "-", "-"]
for (H, W) in [(256, 256)]:
    B, ci, co, k = 16, 64, 64, 3
    x = torch.randn(B, H, W, ci, device="cuda").to(torch.bfloat16)
    w = (torch.randn(co, 9, ci, device="cuda") / 24).to(torch.bfloat16)
    b = torch.randn(co, device="cuda")
    sc = torch.rand(B, ci, device="cuda") + 0.5; sh = torch.randn(B, ci, device="cuda") * 0.1
    res = torch.randn(B, H, W, co, device="cuda").to(torch.bfloat16)
    y = torch.zeros(B, H, W, co, device="cuda", dtype=torch.bfloat16)
    _, st = ops.conv2d_fused(x, w, b, co, k, out=y, stats=True)
    run = lambda: ops.conv2d_fused(x, w, b, co, k, gn=(sc, sh), gn_act=1, res=res, out_scale=0.7071, out=y, stats=st)
    for _ in range(2): run()
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 32)()
    l.diffsep_ws_debug_read(out, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    l.diffsep_ws_debug_read(out, 1)
    print(f"64->64 {H}x{W}: {e0.elapsed_time(e1)/5*1e3:.1f} us/launch")
    for g in range(2):
        nb = out[g * 16 + 15]; tot = sum(out[g * 16 + i] for i in range(12))
        print(f"  group {g}: {tot/nb:.0f} ticks/block")
        for i in range(7):
            print(f"    {names[i]:26s} {out[g*16+i]/nb:9.0f}  {100*out[g*16+i]/tot:5.1f} %")
PY
