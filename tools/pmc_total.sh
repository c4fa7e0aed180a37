#!/bin/bash
# HBM traffic of one whole bench step (all kernels): FETCH_SIZE and WRITE_SIZE in separate passes, kernel-trace only.
# bench.py --in-flight 1 --steps 1 --warmup 0 runs 5 identical steps in total (2 preparation, 1 timed, 2 checks).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/pmc_total}
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_total_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_total_$c -o pmc --output-format csv -- python bench.py --in-flight 1 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-graph > $OUT/$c.log 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
tot, per = {}, {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pmc_total_{c}/**/*counter_collection.csv", recursive=True)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c:
            continue
        v = float(r["Counter_Value"])
        tot[c] = tot.get(c, 0.0) + v
        k = r["Kernel_Name"].split("(")[0][:60]
        per.setdefault(k, {}).setdefault(c, 0.0)
        per[k][c] += v
steps = 5
# guide: FETCH_SIZE / WRITE_SIZE are in KB (1024 B); upper bound 2*FETCH + WRITE (32-byte vs 64-byte request tally), lower FETCH + WRITE
res = {"steps_profiled": steps,
       "hbm_GB_per_step_upper": (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / steps / 1e9,
       "hbm_GB_per_step_lower": (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / steps / 1e9,
       "top_kernels_GB_per_step_upper": {k: round((2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024 / steps / 1e9, 2)
                                         for k, v in sorted(per.items(), key=lambda kv: -(2 * kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0)))[:8]}}
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
PY
