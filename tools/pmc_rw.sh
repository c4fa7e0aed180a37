#!/bin/bash
# SQ counters of the register-weight conv kernel on the stand-alone launch shapes (tools/rw_bench.py): MFMA utilisation,
# wave-state split, instruction counts.  Two --pmc passes (8 SQ slots each), kernel-trace only.  Run via gpurun.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/pmc_rw}
mkdir -p $OUT
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA"
rm -rf /tmp/pmc_rw
rocprofv3 --kernel-trace --pmc $P1 -d /tmp/pmc_rw/p1 -o pmc --output-format csv -- python tools/rw_bench.py 3 > $OUT/run1.log 2>&1
rocprofv3 --kernel-trace --pmc $P2 -d /tmp/pmc_rw/p2 -o pmc --output-format csv -- python tools/rw_bench.py 3 > $OUT/run2.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
res = {}
for f in glob.glob("/tmp/pmc_rw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv3x3_rw_kernel" not in k and "conv3x3_ws1_kernel" not in k and "conv_mfma_kernel" not in k:
            continue
        import re
        m = re.search(r"(conv3x3_rw_kernel<[^>]*>|conv3x3_ws1_kernel<[^>]*>|conv_mfma_kernel<[^>]*>)", k)
        key = (m.group(1) if m else k[:60]) + " grid " + r.get("Grid_Size", "?")
        d = res.setdefault(key, {}).setdefault(r["Counter_Name"], [0.0, 0])
        d[0] += float(r["Counter_Value"]); d[1] += 1
summ = {}
for k, c in sorted(res.items()):
    s = {}
    for name, (v, n) in c.items():
        s[name] = v / n
    s["launches"] = max(n for _, n in c.values())
    if "GRBM_GUI_ACTIVE" in s and "SQ_VALU_MFMA_BUSY_CYCLES" in s:
        g = s["GRBM_GUI_ACTIVE"] / 8
        s["mfma_util"] = s["SQ_VALU_MFMA_BUSY_CYCLES"] / (g * 256 * 4)
        wc = s["SQ_WAVE_CYCLES"]
        for q in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            s[q.lower() + "_frac"] = s[q] / wc
    summ[k] = s
print(json.dumps(summ, indent=1))
json.dump(summ, open(os.path.join(out, "summary.json"), "w"), indent=1)
PY
