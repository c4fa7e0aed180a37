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
import csv, glob, json, os, sys
out = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("OUT", "gpurun_out/pmc_bench")
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(os.path.join("gpurun_out/pmc_bench", c, "*counter_collection.csv"))
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f[0])):
        if "conv_mfma_kernel<unsigned short, 9, 8, 32, 64" in r["Kernel_Name"] and r["Counter_Name"] == c:
            tot += float(r["Counter_Value"]); n += 1
    res[c] = (tot, n)
print(json.dumps(res))
json.dump(res, open("gpurun_out/pmc_bench/summary.json", "w"))
PY
