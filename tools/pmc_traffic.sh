#!/bin/bash
# HBM traffic per kernel instantiation inside the real sampler: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots),
# kernel-trace only, as MI355X_MICROARCH.md prescribes; N = 4 reverse steps (rocprofv3 --pmc crashed on the 60-evaluation
# run in round 2), one batch at a time, eager launches.  The launch mix per evaluation is that of the full run, so the
# per-launch averages of an instantiation are comparable with bench.py's roofline.achieved of the same instantiation.
# Usage (GPU box, repo root): COMMIT=<hash> bash tools/pmc_traffic.sh [outdir] [dtype]; the result goes to
# profiles/pmc_conv3x3.json (bench.py reads roofline.traffic from there and names the commit).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/pmc_traffic}
DT=${2:-f16}
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_traffic/$c -o pmc --output-format csv -- python bench.py --dtype $DT --steps 1 --warmup 0 -N 4 --in-flight 1 --no-cpu-baseline --no-roofline --no-graph --no-extra-modes > $OUT/$c.log 2>&1
done
OUT=$OUT DT=$DT python - <<'PY'
import csv, glob, json, os, re
out, dt = os.environ["OUT"], os.environ["DT"]
def short(k):
    m = re.search(r"((?:conv3x3_rw|conv3x3_sws|conv3x3_sw|conv3x3_ws1|conv_mfma|conv3x3_small|conv3x3_thin_in|conv3x3_thin_out|attn_fused)_kernel(?:<[^>]*>)?)", k)
    return m.group(1).replace("unsigned short", dt if dt in ("f16", "bf16") else "bf16").replace("float", "f32").replace(" ", "") if m else None
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"/tmp/pmc_traffic/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k and r["Counter_Name"] == c:
                d = res.setdefault(k, {}).setdefault(c, [0.0, 0])
                d[0] += float(r["Counter_Value"]); d[1] += 1
kern = {}
for k, v in res.items():
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    n = v["FETCH_SIZE"][1]
    f, w = v["FETCH_SIZE"][0] / n, v["WRITE_SIZE"][0] / max(v["WRITE_SIZE"][1], 1)
    kern[k] = {"launches": n, "fetch_size_kb_per_launch": round(f, 1), "write_size_kb_per_launch": round(w, 1),
               "hbm_bytes_per_launch_guide_formula": round((2 * f + w) * 1024), "hbm_bytes_per_launch_lower_bound": round((f + w) * 1024)}
kern = dict(sorted(kern.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch_guide_formula"] * kv[1]["launches"]))
doc = {"source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --dtype {dt} --steps 1 "
                 "--warmup 0 -N 4 --in-flight 1 --no-graph` (tools/pmc_traffic.sh; with N = 30 rocprofv3 --pmc crashed in one of its own threads)",
       "commit": os.environ.get("COMMIT", "unrecorded"), "dtype": dt,
       "correction": "MI355X_MICROARCH.md (HBM): FETCH_SIZE tallies 128-byte requests at 64 B -> doubled for wide coalesced reads; "
                     "WRITE_SIZE uncorrected; unit KB = 1024 B.  2*FETCH+WRITE is an upper bound, FETCH+WRITE a lower bound of the true "
                     "traffic (calibration: DESIGN.md section 5).  Per launch, averaged over every launch of the instantiation (all shapes).",
       "by_instantiation": kern}
json.dump(doc, open(os.path.join(out, "pmc_conv3x3.json"), "w"), indent=1)
for k, v in list(kern.items())[:8]:
    print(k, v)
PY
