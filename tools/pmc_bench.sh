#!/bin/bash
# HBM traffic of the dominant kernel inside the real bench: FETCH_SIZE and WRITE_SIZE in separate passes
# (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2), kernel-trace only, as MI355X_MICROARCH.md prescribes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/pmc_bench}
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $OUT/$c -o pmc --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-graph > $OUT/$c.log 2>&1
done
python - <<'PY'
import csv, glob, json, os
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(os.path.join("gpurun_out/pmc_bench", c, "*counter_collection.csv"))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        key = ("conv_mfma_9_8_32_64" if "conv_mfma_kernel<unsigned short, 9, 8, 32, 64" in k else
               "conv3x3_ws1" if "conv3x3_ws1_kernel" in k else None)
        if key and r["Counter_Name"] == c:
            d = res.setdefault(key, {}).setdefault(c, [0.0, 0])
            d[0] += float(r["Counter_Value"]); d[1] += 1
print(json.dumps(res))
json.dump(res, open("gpurun_out/pmc_bench/summary.json", "w"))
PY
