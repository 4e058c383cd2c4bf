#!/bin/bash
# GPU box: the dense-search core micro-benchmark at the candidate counts scripts/sim_dense.py finds for C3 (budget 16 groups per tile; median
# 27; mean 45), then its counters at 27 blocks per tile.  Output: gpurun_out/dense_core.txt
ROOT=/root/repo
OUT=$ROOT/gpurun_out/dense_core.txt
mkdir -p $ROOT/gpurun_out
: > $OUT
for b in 8 16 27 45; do $ROOT/build/dense_core 15625 $b | tee -a $OUT; done
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
  rm -rf /tmp/dc_pmc
  timeout -s KILL 100 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/dc_pmc -o dc -- $ROOT/build/dense_core 15625 27 > /tmp/dc_pmc.log 2>&1 || tail -3 /tmp/dc_pmc.log
  python3 - <<'PY' | tee -a $OUT
import csv, glob
agg = {}
for f in glob.glob('/tmp/dc_pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'dense_core' in r.get('Kernel_Name', ''):
            agg.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
print("counters per launch (27 blocks per tile, 15625 tiles):", {k: round(sum(v) / len(v)) for k, v in agg.items()})
PY
done
