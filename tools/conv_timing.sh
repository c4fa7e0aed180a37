#!/bin/bash
# Build a profiling variant of the library with per-phase cycle counters in the conv kernel and print the
# average cycles per block and phase for a few layer shapes.  (Run on the GPU box via gpurun.)
set -e
cd $(dirname $0)/../diffusion-separation_amd/csrc
mkdir -p ../abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -DCONV_TIMING $CONV_EXTRA -c conv_mfma.hip -o /tmp/conv_timing.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/lib_timing.so /tmp/conv_timing.o $(ls build/*.o | grep -Ev '/(conv_mfma\.o)$')
cd ../..
DIFFSEP_LIB=$PWD/diffusion-separation_amd/abl/lib_timing.so python - <<'PY'
import ctypes, sys, os, torch
sys.path.insert(0, "diffusion-separation_amd")
from diffsep_amd import ops, _lib
l = ctypes.CDLL(os.environ["DIFFSEP_LIB"])
names = ["setup (offsets, descriptors)", "issue first loads", "wait first loads", "activation of chunk 0", "barrier+LDS store+barrier", "issue next loads", "MFMA loop (+activation of next)", "acc dump+barriers+residual loads", "epilogue LDS read+math", "epilogue global stores", "statistics reduce", "-"]
# CONV_CASES="k,cin,cout,H,W;..."; CONV_OPTS="name=1,..." (process options, e.g. no_rw=1,kc64=1)
CASES = [tuple(int(v) for v in c.split(",")) for c in os.environ["CONV_CASES"].split(";")] if os.environ.get("CONV_CASES") else \
    [(3, 128, 64, 256, 256), (3, 128, 128, 64, 64), (3, 256, 128, 32, 32)]
for kv in [v for v in os.environ.get("CONV_OPTS", "").split(",") if v]:
    _lib.check(_lib.lib().diffsep_set_option(kv.split("=")[0].encode(), int(kv.split("=")[1])))
for (k, ci, co, H, W) in CASES:
    B = 16
    x = torch.randn(B, H, W, ci, device="cuda").to(torch.bfloat16)
    w = (torch.randn(co, k * k, ci, device="cuda") / (k * k * ci) ** 0.5).to(torch.bfloat16)
    b = torch.randn(co, device="cuda")
    sc = torch.rand(B, ci, device="cuda") + 0.5; sh = torch.randn(B, ci, device="cuda") * 0.1
    res = torch.randn(B, H, W, co, device="cuda").to(torch.bfloat16)
    run = lambda: ops.conv2d_fused(x, w, b, co, k, gn=(sc, sh), gn_act=1, res=res, out_scale=0.7071)
    for _ in range(2): run()
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    l.diffsep_debug_read(out, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    l.diffsep_debug_read(out, 1)
    nb = out[15]
    tot = sum(out[i] for i in range(12))
    print(f"k{k} {ci}->{co} {H}x{W}: {e0.elapsed_time(e1)/5*1e3:.1f} us/launch, {nb//5} blocks, {tot/nb:.0f} cycles/block ")
    for i in range(11):
        print(f"    {names[i]:28s} {out[i]/nb:9.0f}  {100*out[i]/tot:5.1f} %")
PY
