#!/bin/bash
# kernel trace of the C++ flow driver (examples/odometry_benchmark_flow.cpp): GPU busy time (union of the kernel intervals), summed kernel
# time, how many kernels run at once, and the idle gaps by neighbouring kernels.  usage: odom_flow_trace.sh P R [frames]
ROOT=$PWD
P=${1:-2}; R=${2:-2}; N=${3:-60}
OUT=$ROOT/gpurun_out/flow_trace
rm -rf "$OUT"; mkdir -p "$OUT"
python - <<PY
import sys; sys.path.insert(0, "$ROOT")
from small_gicp_amd import odometry
exe, data = odometry._cpp_driver("odometry_benchmark_flow.cpp", "$OUT", $N)
print(exe, data)
PY
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/prof" -o t -- "$OUT/odometry_benchmark_flow" "$OUT/velodyne" "$OUT/traj.txt" --max_frames $N --preprocess_workers $P --registration_workers $R --repeat 3 ${PINFLAG} > "$OUT/log.txt" 2>&1
cd "$ROOT"
grep "^run=" "$OUT/log.txt"
python - "$(find $OUT/prof -name '*kernel_trace.csv' | head -1)" "$(grep '^run=2' $OUT/log.txt | sed 's/.*window_ns=//')" $N <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
t0, t1 = (int(float(x)) for x in sys.argv[2].split(","))
n = int(sys.argv[3])
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows if t0 <= int(r["Start_Timestamp"]) <= t1)
busy, cur_s, cur_e, total = 0, None, None, 0
for s, e, _, _ in iv:
    total += e - s
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("flow: wall %.1f us/scan; kernels summed %.1f us/scan; GPU busy (union) %.1f us/scan; idle %.1f us/scan; launches %.1f /scan; queues %s" % ((t1 - t0) / 1e3 / n, total / 1e3 / n, busy / 1e3 / n, ((t1 - t0) - busy) / 1e3 / n, len(iv) / n, sorted(set(q for _, _, _, q in iv))))
# time with k kernels in flight
ev = sorted([(s, 1) for s, e, _, _ in iv] + [(e, -1) for s, e, _, _ in iv])
depth, last, hist = 0, t0, collections.Counter()
for t, d in ev:
    hist[depth] += t - last
    last, depth = t, depth + d
hist[0] += t1 - last
print("time with k kernels in flight, us/scan: " + "  ".join("%d: %.1f" % (k, v / 1e3 / n) for k, v in sorted(hist.items())))
by = collections.Counter(); cnt = collections.Counter()
for s, e, name, _ in iv:
    by[name.split("(")[0][-60:]] += e - s; cnt[name.split("(")[0][-60:]] += 1
print("kernels, us/scan (calls/scan):")
for k, v in by.most_common(16):
    print("  %7.1f (%4.1f)  %s" % (v / 1e3 / n, cnt[k] / n, k))
gaps, cur = [], None
for s, e, name, q in iv:
    if cur is not None and s > cur[0]: gaps.append((s - cur[0], cur[1], name))
    if cur is None or e > cur[0]: cur = (e, name)
agg = collections.Counter()
for g, a, b in gaps: agg[(a.split("(")[0][-40:], b.split("(")[0][-40:])] += g
print("idle time by (kernel before -> kernel after), us/scan:")
for (a, b), g in agg.most_common(10): print("  %7.1f  %s -> %s" % (g / 1e3 / n, a, b))
PY
rm -rf "$OUT"
