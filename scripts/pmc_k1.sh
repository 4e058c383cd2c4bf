#!/bin/bash
# PMC deep-dive of the linearize kernel on the bench workload (run on the GPU box). Usage: pmc_k1.sh <tag>
set -u
TAG=${1:-k1}
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline"
rocprofv3 -L > "$OUT/counters.txt" 2>&1
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA" \
           "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -o pmc -- $BENCH > "$OUT/p$i.log" 2>&1
done
cd "$ROOT"
python - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
out = sys.argv[1]
agg = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r.get("Kernel_Name", "")
        if "linearize_kernel" not in kn and "error_kernel" not in kn: continue
        k = ("K1" if "linearize" in kn else "K2", r.get("Counter_Name"))
        agg[k][0] += float(r.get("Counter_Value", 0)); agg[k][1] += 1
for (kn, cn), (v, c) in sorted(agg.items()):
    print("%s %-32s per_launch=%.4g (n=%d)" % (kn, cn, v / max(c, 1), c))
PY
