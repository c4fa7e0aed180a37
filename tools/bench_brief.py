#!/usr/bin/env python3
"""bench.py (main dtype only) reduced to the lines an A/B needs: value, one batch alone, the dominant kernel, selected shapes and
HBM-bound kernels.  Usage: python tools/bench_brief.py [substring of shape / kernel names to list] -- the library variants come from
DIFFSEP_LIB_F16 / DIFFSEP_* environment variables."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1:] or ["thin_out", "128->64 @256", "64->64 @256", "@128x128"]
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-extra-modes", "--no-cpu-baseline"], capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not line:
    print(out.stderr[-800:]); sys.exit(1)
r = json.loads(line[-1])
ro = r["roofline"]
print(f"value {r['value']:.2f} utt/s  ms_per_step {r['ms_per_step']:.1f}  one batch alone {r['one_batch_alone_ms']:.1f} ms  dominant {ro['kernel']} {ro['avg_launch_us']:.1f} us frac {ro['frac']:.3f}  "
      f"all MFMA kernels {ro['all_mfma_kernels_ms']:.1f} ms  HBM kernels {ro['hbm_kernels_total_ms']:.1f} ms")
for s_ in ro["per_shape"]:
    if any(p in s_["shape"] or p in s_["kernel"] for p in pat):
        print(f"   {s_['kernel'][:44]:44s} {s_['shape']:34s} x{s_['launches']:4d} {s_['avg_us']:7.1f} us")
for h in ro["hbm_kernels"][:6]:
    print(f"   [hbm] {h['kernel'][:40]:40s} {h['shape']:16s} {h['avg_us']:7.1f} us  {h['frac_hbm']:.3f}")
