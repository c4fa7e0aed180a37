#!/bin/bash
# Memory-path PMC passes over the conv micro-benchmark (one counter group per run, kernel-trace only).
# usage: tools/pmc_mem.sh <shape-index-list> <outdir>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SEL=${1:-0}; OUT=${2:-gpurun_out/pmc_mem}
mkdir -p $OUT
i=0
for grp in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_avr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum" \
           "SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE GRBM_TA_BUSY GRBM_TC_BUSY GRBM_EA_BUSY GRBM_UTCL2_BUSY"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $grp -d $OUT/p$i -o pmc --output-format csv -- python tools/bench_conv.py bf16 5 $SEL > $OUT/p$i.log 2>&1 || echo "group $i failed: $grp"
done
python - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob("$OUT/p*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv" in k and "at::" not in k:
            acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print("==", k)
    for c, v in sorted(cs.items()):
        print(f"   {c:40s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
PY
