import ctypes, sys, os, torch
sys.path.insert(0, "diffusion-separation_amd")
from diffsep_amd import ops, _lib
l = ctypes.CDLL(os.environ["DIFFSEP_LIB"])
names = ["setup (offsets, descriptors)", "issue first loads", "wait first loads", "activation of chunk 0", "barrier+LDS store+barrier", "issue next loads", "MFMA loop (+activation of next)", "acc dump+barriers+residual loads", "epilogue LDS read+math", "epilogue global stores", "statistics reduce", "-"]
for (k, ci, co, H, W, split) in [(3, 64, 64, 256, 256, True), (3, 64, 64, 256, 256, False), (3, 128, 128, 64, 64, True)]:
    B = 16
    x = torch.randn(B, H, W, ci, device="cuda")
    w = ops.pack_conv_weight(torch.randn(co, ci, k, k) / (k * k * ci) ** 0.5, torch.float32, chunk=16).cuda()
    b = torch.randn(co, device="cuda")
    sc = torch.rand(B, ci, device="cuda") + 0.5; sh = torch.randn(B, ci, device="cuda") * 0.1
    res = torch.randn(B, H, W, co, device="cuda")
    y = torch.zeros(B, H, W, co, device="cuda")
    run = lambda: ops.conv2d_fused(x, w, b, co, k, gn=(sc, sh), gn_act=1, res=res, out_scale=0.7071, out=y, w_chunk=16, split=split)
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
    print(f"k{k} {ci}->{co} {H}x{W} split={split}: {e0.elapsed_time(e1)/5*1e3:.1f} us/launch, {nb//5} blocks, {tot/nb:.0f} cycles/block ")
    for i in range(11):
        print(f"    {names[i]:28s} {out[i]/nb:9.0f}  {100*out[i]/tot:5.1f} %")
