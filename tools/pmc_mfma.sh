#!/bin/bash
# MFMA utilisation and wave-state split of the dominant kernels inside the real bench (one batch at a time, eager
# launches): one SQ + GRBM pass, kernel-trace only.  MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs
# * 4 SIMDs) (the gfx94x formula; GRBM_GUI_ACTIVE comes summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES = 32 per
# v_mfma_f32_32x32x16_bf16).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/pmc_mfma}
mkdir -p $OUT
CTRS="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/pmc_mfma -o pmc --output-format csv -- python bench.py --in-flight 1 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-graph > $OUT/run.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
f = glob.glob("/tmp/pmc_mfma/**/*counter_collection.csv", recursive=True)
res = {}
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    key = ("conv_mfma<bf16,9,8,32,64>" if "conv_mfma_kernel<unsigned short, 9, 8, 32, 64" in k else
           "conv3x3_ws1" if "conv3x3_ws1_kernel" in k else
           "conv_mfma<bf16,9,8,8,64>" if "conv_mfma_kernel<unsigned short, 9, 8, 8, 64" in k else None)
    if key:
        d = res.setdefault(key, {}).setdefault(r["Counter_Name"], [0.0, 0])
        d[0] += float(r["Counter_Value"]); d[1] += 1
summ = {}
for k, c in res.items():
    n = c["GRBM_GUI_ACTIVE"][1]
    g = c["GRBM_GUI_ACTIVE"][0] / 8
    s = {"launches": n, "gui_active_cycles_per_launch": g / n,
         "mfma_util": c["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (g * 256 * 4)}
    wc = c["SQ_WAVE_CYCLES"][0]
    for q in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        s[q.lower() + "_frac_of_wave_cycles"] = c[q][0] / wc
    summ[k] = s
print(json.dumps(summ, indent=1))
json.dump({"raw": res, "summary": summ}, open(os.path.join(out, "summary.json"), "w"), indent=1)
PY
