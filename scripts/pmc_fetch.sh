#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per launch of the K1 kernels for the environment given as arguments (diagnostics):  pmc_fetch.sh "A=1" "A=0"
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pf; env $cfg timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pf -o pmc -- python $ROOT/scripts/diag_passes.py > /dev/null 2>&1
    python - "$cfg" $ctr <<'PY'
import csv, glob, sys
from collections import defaultdict
agg = defaultdict(lambda: [0.0, 0])
for f in glob.glob("/tmp/pf/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != sys.argv[2]: continue
        kn = r.get("Kernel_Name", "")
        for key in ("search_linearize_kernel<float, 2, false>", "search_linearize_kernel<float, 2, true>", "nn_search_queue_kernel", "reduce_rows_kernel"):
            if key in kn:
                agg[key][0] += float(r["Counter_Value"]); agg[key][1] += 1
print(sys.argv[1], sys.argv[2], {k: "%.1f MB x%d" % (v / c / 1024 * (2 if sys.argv[2] == "FETCH_SIZE" else 1), c) for k, (v, c) in agg.items()})
PY
  done
done
