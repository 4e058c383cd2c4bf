#!/bin/bash
# durations of the individual kd_split_level_kernel / kd_finish dispatches of a few scans (diagnostics)
ROOT=/root/repo
rm -rf $ROOT/gpurun_out/split_trace; mkdir -p $ROOT/gpurun_out/split_trace
cd /tmp && export TMPDIR=/tmp
cat > /tmp/odom_run3.py <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from small_gicp_amd import odometry
odometry.run_synthetic(3)
PY
timeout -s KILL 150 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/split_trace -o t -- python /tmp/odom_run3.py > /dev/null 2>&1
cd $ROOT
t=$(find gpurun_out/split_trace -name "*kernel_trace.csv" | head -1)
python - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
for r in rows[-75:]:
    n = r["Kernel_Name"].split("(")[0][-48:]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print("%7.1f us (gap %6.1f) grid %6s  %s" % ((e - s) / 1e3, gap, r.get("Grid_Size_X", r.get("Grid_Size", "?")), n))
    prev_end = e
PY
rm -rf $ROOT/gpurun_out/split_trace
