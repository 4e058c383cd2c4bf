#!/bin/bash
# kernel trace of the pipelined C5 driver: GPU busy time (union of the kernel intervals), summed kernel time and idle time per scan
ROOT=$PWD
OUT=$ROOT/gpurun_out/pipe_trace
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cat > /tmp/pipe_run.py <<PY
import sys; sys.path.insert(0, "$ROOT")
from small_gicp_amd import odometry, synthetic
import time
scans = [synthetic.kitti_like_scan(f)[0] for f in range(40)]
od = odometry.PipelinedOdometry(workers=${WORKERS:-2}) if ${WORKERS:-2} > 0 else None
class Seq:
    def run(self, scans):
        import time as _t
        o = odometry.OnlineOdometry(); t0 = _t.perf_counter(); poses = [o.estimate(s) for s in scans]; return poses, _t.perf_counter() - t0, []
od = od or Seq()
od.run(scans[:3])
t0 = time.clock_gettime_ns(time.CLOCK_MONOTONIC)
poses, wall, iters = od.run(scans)
t1 = time.clock_gettime_ns(time.CLOCK_MONOTONIC)
print("WINDOW %d %d %.3f" % (t0, t1, 1e3 * wall / len(scans)))
PY
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o t -- python /tmp/pipe_run.py > "$OUT/log.txt" 2>&1
cd "$ROOT"
grep WINDOW "$OUT/log.txt"
python - "$(find $OUT -name '*kernel_trace.csv' | head -1)" "$(grep WINDOW $OUT/log.txt)" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
_, t0, t1, ms = sys.argv[2].split()
t0, t1 = int(t0), int(t1)
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
n = 40
print("pipelined: wall %.1f us/scan; kernels summed %.1f us/scan; GPU busy (union) %.1f us/scan; idle %.1f us/scan; launches %.1f /scan; queues %s" % ((t1 - t0) / 1e3 / n, total / 1e3 / n, busy / 1e3 / n, ((t1 - t0) - busy) / 1e3 / n, len(iv) / n, sorted(set(q for _, _, _, q in iv))))
# the longest idle gaps: what ran before and after
gaps = []
cur_e = None
for s, e, name, q in iv:
    if cur_e is not None and s > cur_e[0]:
        gaps.append((s - cur_e[0], cur_e[1], name))
    if cur_e is None or e > cur_e[0]:
        cur_e = (e, name)
import collections
agg = collections.Counter()
for g, a, b in gaps:
    agg[(a.split("(")[0][-40:], b.split("(")[0][-40:])] += g
print("idle time by (kernel before -> kernel after), us/scan:")
for (a, b), g in agg.most_common(14):
    print("  %7.1f  %s -> %s" % (g / 1e3 / n, a, b))
PY
rm -rf "$OUT"
