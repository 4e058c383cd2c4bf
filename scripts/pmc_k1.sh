#!/bin/bash
# PMC deep-dive of the linearize kernel on the bench workload (run on the GPU box). Usage: pmc_k1.sh <tag>
set -u
TAG=${1:-k1}
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
BENCH=${PMC_CMD:-"python $ROOT/scripts/diag_passes.py"}
rocprofv3 -L > "$OUT/counters.txt" 2>&1
i=0
# PMC_EXTRA=1 adds the cache / TA / LDS passes
# always: the clock the kernels ran at (GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles / 8 / duration)
for set in "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA" ${PMC_EXTRA:+"TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TA_TA_BUSY_sum TD_TD_BUSY_sum" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT" "GRBM_GUI_ACTIVE GRBM_COUNT"}; do
  i=$((i+1))
  timeout -s KILL 100 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -o pmc -- $BENCH > "$OUT/p$i.log" 2>&1
done
cd "$ROOT"
python - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
out = sys.argv[1]
def name_of(kn):
    if "grid_ring1" in kn: return "G_ring1"      # cell grid: ring 1 of every query
    if "grid_finish" in kn: return "G_finish"    # cell grid: the queries ring 1 left open
    if "certify_linearize" in kn: return "K1_certify"  # warm pass: certificate check inside the streaming factor kernel + walk phase
    if "search_linearize" in kn: return "K1_lane_warm" if "true" in kn else "K1_lane_cold"   # search + factors, one query per lane
    if "nn_search_queue" in kn: return "K1_queue_warm"                                        # check + queue-fed walks (+ factors)
    if "nn_search" in kn: return "K1a_warm" if "true" in kn else "K1a_cold"                   # search only (non-fused paths)
    return "K1b" if "linearize" in kn else "K2"
agg = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r.get("Kernel_Name", "")
        if "linearize_kernel" not in kn and "error_kernel" not in kn and "nn_search" not in kn and "grid_" not in kn: continue  # (search_linearize_kernel matches too)
        k = (name_of(kn), r.get("Counter_Name"))
        agg[k][0] += float(r.get("Counter_Value", 0)); agg[k][1] += 1
dur = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(out, "p1", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r.get("Kernel_Name", "")
        if "linearize_kernel" in kn or "error_kernel" in kn or "nn_search" in kn or "grid_" in kn:
            key = name_of(kn)
            dur[key][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; dur[key][1] += 1
for k, (v, c) in sorted(dur.items()):
    print("%s avg_us=%.2f (n=%d)" % (k, v / max(c, 1), c))
for (kn, cn), (v, c) in sorted(agg.items()):
    print("%s %-32s per_launch=%.4g (n=%d)" % (kn, cn, v / max(c, 1), c))
PY
rm -rf "$OUT"/p*/  # raw per-dispatch tables are large; only the aggregate above travels back
