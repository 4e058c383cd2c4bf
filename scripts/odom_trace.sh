#!/bin/bash
# kernel trace of the scan-to-scan odometry leg (C5): launches per scan, kernel time per scan, top kernels
ROOT=/root/repo
mkdir -p $ROOT/gpurun_out/odom_trace; rm -rf $ROOT/gpurun_out/odom_trace/*
cd /tmp && export TMPDIR=/tmp
cat > /tmp/odom_run.py <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from small_gicp_amd import odometry
r = odometry.run_synthetic(10)
print({k: v for k, v in r.items() if k not in ("estimated", "ground_truth")})
PY
timeout -s KILL 60 python /tmp/odom_run.py
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/odom_trace -o t -- python /tmp/odom_run.py > $ROOT/gpurun_out/odom_trace/log.txt 2>&1
cd $ROOT
t=$(find gpurun_out/odom_trace -name "*kernel_trace.csv" | head -1)
python - "$t" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print("dispatches", len(rows), "-> per scan (10 scans)", len(rows) / 10)
tot = collections.Counter(); cnt = collections.Counter()
for r in rows:
    n = r["Kernel_Name"].split("(")[0][-60:]
    tot[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; cnt[n] += 1
print("kernel time per scan %.1f us" % (sum(tot.values()) / 10))
for n, v in tot.most_common(22):
    print("%8.1f us/scan %5.1f calls/scan  %s" % (v / 10, cnt[n] / 10, n))
PY
rm -f $t
