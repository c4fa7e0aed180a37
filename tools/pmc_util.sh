#!/bin/bash
# Counters of every MFMA kernel inside the real sampler (one batch at a time, eager launches, N = 4 reverse steps:
# rocprofv3 --pmc on the full 60-evaluation run crashed in round 2): MFMA utilisation, the effective shader clock
# (GRBM_GUI_ACTIVE / 8 XCDs / kernel duration from the same pass's kernel trace), instructions per MFMA, wave states.
# Two --pmc passes (SQ slots), kernel-trace only.  Run via gpurun; writes $OUT/summary.json.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/pmc_r03}
DT=${2:-f16}
mkdir -p $OUT
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA"
CMD="python bench.py --dtype $DT --in-flight 1 --steps 1 --warmup 0 -N 4 --no-cpu-baseline --no-roofline --no-graph --no-extra-modes"
rm -rf /tmp/pmc_r03
rocprofv3 --kernel-trace --pmc $P1 -d /tmp/pmc_r03/p1 -o pmc --output-format csv -- $CMD > $OUT/run1.log 2>&1
rocprofv3 --kernel-trace --pmc $P2 -d /tmp/pmc_r03/p2 -o pmc --output-format csv -- $CMD > $OUT/run2.log 2>&1
DT=$DT python - "$OUT" "$CMD" <<'PY'
import csv, glob, json, os, re, sys
out, cmd = sys.argv[1], sys.argv[2]
def short(k):
    m = re.search(r"((?:conv3x3_rw|conv3x3_sws|conv3x3_sw|conv3x3_ws1|conv_mfma|conv3x3_small|conv3x3_thin_in|conv3x3_thin_out|attn_fused)_kernel(?:<[^>]*>)?)", k)
    return m.group(1).replace("unsigned short", os.environ.get("DT", "bf16")).replace("float", "f32").replace(" ", "") if m else None
res, dur = {}, {}
for p in ("p1", "p2"):
    disp = {}
    for f in glob.glob(f"/tmp/pmc_r03/{p}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            disp[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for f in glob.glob(f"/tmp/pmc_r03/{p}/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not k:
                continue
            key = f"{k} grid {r.get('Grid_Size', '?')}"
            d = res.setdefault(key, {}).setdefault(r["Counter_Name"], [0.0, 0])
            d[0] += float(r["Counter_Value"]); d[1] += 1
            if p == "p1" and r["Dispatch_Id"] not in seen and r["Dispatch_Id"] in disp:
                seen.add(r["Dispatch_Id"])
                a = dur.setdefault(key, [0, 0]); a[0] += disp[r["Dispatch_Id"]]; a[1] += 1
summ = {}
for k, c in res.items():
    s = {n: v / m for n, (v, m) in c.items()}
    o = {"launches": max(m for _, m in c.values())}
    if k in dur:
        o["avg_us_under_counters"] = round(dur[k][0] / dur[k][1] / 1e3, 1)
    if "GRBM_GUI_ACTIVE" in s:
        g = s["GRBM_GUI_ACTIVE"] / 8
        o["gui_active_cycles"] = round(g)
        # GRBM_GUI_ACTIVE keeps counting around a short dispatch (round 4 printed "clocks" of 3 - 5 GHz for kernels under ~30 us): the
        # clock estimate and the GRBM-based utilisation are given for dispatches of >= 50 us only
        longish = k in dur and dur[k][0] / dur[k][1] >= 50e3
        if longish:
            o["shader_clock_ghz"] = round(g / (dur[k][0] / dur[k][1]), 3)
            o["mfma_util"] = round(s["SQ_VALU_MFMA_BUSY_CYCLES"] / (g * 256 * 4), 4)
        wc = s["SQ_WAVE_CYCLES"]
        # independent of GRBM and of the dispatch length: MFMA-busy cycles per wave cycle (SQ_WAVE_CYCLES counts quad-cycles) —
        # the utilisation of the wave's SIMD for the one-wave-per-SIMD kernels (conv3x3_rw); a lower bound with more waves per SIMD
        o["mfma_busy_per_wave_cycle"] = round(s["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * wc), 4)
        o["wave_parked_waitcnt_or_barrier"] = round(s["SQ_WAIT_ANY"] / wc, 3)
        o["wave_issue_stalled"] = round(s["SQ_WAIT_INST_ANY"] / wc, 3)
        o["wave_issuing"] = round(s["SQ_ACTIVE_INST_ANY"] / wc, 3)
    if s.get("SQ_INSTS_MFMA"):
        m = s["SQ_INSTS_MFMA"]
        o["valu_per_mfma"] = round((s.get("SQ_INSTS_VALU", 0) - m) / m, 2) if s.get("SQ_INSTS_VALU") else None
        o["lds_per_mfma"] = round(s["SQ_INSTS_LDS"] / m, 2)
        o["salu_per_mfma"] = round(s["SQ_INSTS_SALU"] / m, 2)
        o["vmem_per_mfma"] = round((s["SQ_INSTS_VMEM_RD"] + s["SQ_INSTS_VMEM_WR"]) / m, 3)
        o["lds_bank_conflict_cycles_per_lds_inst"] = round(s["SQ_LDS_BANK_CONFLICT"] / max(s["SQ_INSTS_LDS"], 1), 3)
    summ[k] = o
summ = dict(sorted(summ.items(), key=lambda kv: -kv[1].get("gui_active_cycles", 0) * kv[1]["launches"]))
doc = {"commit": os.environ.get("COMMIT", "unrecorded"), "dtype": os.environ.get("DT", "bf16"),
       "command": "rocprofv3 --kernel-trace --pmc <8 counters> -- " + cmd + "  (two passes, tools/pmc_util.sh)",
       "note": "GRBM_GUI_ACTIVE is summed over the 8 XCDs; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs): per "
               "shader CYCLE, whatever the clock; shader_clock_ghz = GRBM_GUI_ACTIVE / 8 / kernel duration of the same dispatches (both only for dispatches >= 50 us); "
               "mfma_busy_per_wave_cycle = SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_WAVE_CYCLES); "
               "SQ_INSTS_VALU includes the MFMAs (subtracted in valu_per_mfma); wave states as fractions of SQ_WAVE_CYCLES",
       "kernels": summ}
json.dump(doc, open(os.path.join(out, "summary.json"), "w"), indent=1)
for k, o in list(summ.items())[:14]:
    print(k, o)
PY
