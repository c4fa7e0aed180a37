#!/bin/bash
# Per-phase cycle counters of the register-weight conv kernel (profiling build: -DRW_TIMING).  Run via gpurun.
set -e
cd $(dirname $0)/../diffusion-separation_amd/csrc
mkdir -p ../abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=1000000 -DRW_TIMING -DDS_HALF_F16 $RW_EXTRA -c conv3x3_rw.hip -o /tmp/rw_timing.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/lib_rwtiming.so /tmp/rw_timing.o $(ls build_f16/*.o | grep -Ev '/(conv3x3_rw\.o)$')
cd ../..
DIFFSEP_LIB_F16=$PWD/diffusion-separation_amd/abl/lib_rwtiming.so python - <<'PY'
import ctypes, sys, os, torch
sys.path.insert(0, "diffusion-separation_amd")
from diffsep_amd import ops
l = ctypes.CDLL(os.environ["DIFFSEP_LIB_F16"])
names = ["prologue: staging of the first chunk", "barrier wait", "3x3 chunk phases", "skip chunk phases", "epilogue", "tail (statistics)", "prologue: tables + descriptors", "prologue: first loads + weight fragments issued, barrier"]
import os
BS = [int(v) for v in os.environ.get('RW_B', '16').split(',')]
# RW_CASES="cin,cout,H,W,fused;..." (default: the 64-cout shapes of the large levels)
CASES = [tuple(int(v) for v in c.split(",")) for c in os.environ["RW_CASES"].split(";")] if os.environ.get("RW_CASES") else \
    [(64, 64, 256, 256, 2), (64, 64, 256, 256, 1), (64, 64, 256, 256, 0), (128, 64, 256, 256, 1), (128, 64, 256, 256, 0), (64, 64, 128, 128, 1)]
from diffsep_amd import _lib
_lib.lib().diffsep_set_option(b"rw_small", 1)
for (ci, co, H, W, fused, B) in [(c, o, hh, ww, f, bb) for bb in BS for (c, o, hh, ww, f) in CASES]:
    k = 3
    x = torch.randn(B, H, W, ci, device="cuda").to(torch.float16)
    w = (torch.randn(co, 9, ci, device="cuda") / (9 * ci) ** 0.5).to(torch.float16)
    kc = 32 if os.environ.get("RW_CHUNKED", "1") == "1" else 0  # chunk-major [Cin/32][9][Cout][32] like the engine's weights
    wkw = {}
    if kc:
        w = w.reshape(co, 9, ci // kc, kc).permute(2, 1, 0, 3).contiguous()
        wkw = dict(w_chunk=kc)
    b = torch.randn(co, device="cuda")
    sc = torch.rand(B, ci, device="cuda") + 0.5; sh = torch.randn(B, ci, device="cuda") * 0.1
    res = torch.randn(B, H, W, co, device="cuda").to(torch.float16)
    y = torch.zeros(B, H, W, co, device="cuda", dtype=torch.float16)
    _, st = ops.conv2d_fused(x, w, b, co, k, out=y, stats=True, **wkw)
    if fused == 2:    # Conv_1 of a plain block: GroupNorm + SiLU, bias, residual, 1/sqrt(2), statistics
        run = lambda: ops.conv2d_fused(x, w, b, co, k, gn=(sc, sh), gn_act=1, res=res, out_scale=0.7071, out=y, stats=st, **wkw)
    elif fused == 1:  # Conv_0: GroupNorm + SiLU, bias, statistics
        run = lambda: ops.conv2d_fused(x, w, b, co, k, gn=(sc, sh), gn_act=1, out=y, stats=st, **wkw)
    else:
        run = lambda: ops.conv2d_fused(x, w, b, co, k, out=y, **wkw)
    for _ in range(2): run()
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    l.diffsep_rw_debug_read(out, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    l.diffsep_rw_debug_read(out, 1)
    nb = out[15]; tot = sum(out[i] for i in range(8))
    if nb == 0:
        continue  # (this launch did not run on the register-weight kernel)
    print(f"B={B} {ci}->{co} {H}x{W} {['plain', 'GN+SiLU+stats', 'GN+SiLU+stats+residual'][fused]}: {e0.elapsed_time(e1)/5*1e3:.1f} us/launch, {tot/nb:.0f} ticks/block (wave 0), {nb//5} blocks -> {tot/nb/(e0.elapsed_time(e1)/5*1e3)/1e3:.2f} ticks/ns")
    for i in range(8):
        print(f"    {names[i]:36s} {out[i]/nb:9.0f}  {100*out[i]/tot:5.1f} %")
PY
