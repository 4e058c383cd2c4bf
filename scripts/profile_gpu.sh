#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash scripts/profile_gpu.sh <tag> [steps]
# Produces gpurun_out/prof_<tag>/: rocprofv3 kernel-trace stats of the bench workload, FETCH_SIZE / WRITE_SIZE in separate PMC passes
# (kernel-trace only: no sys/hip/hsa tracing together with --pmc), a kernel trace of the odometry leg, and the condensed files
# <tag>_rocprofv3_summary.txt/.json, <tag>_kernel_stats.csv, <tag>_odom_kernel_stats.csv, <tag>_k1_traffic.json, <tag>_odom_trace.json
# which the builder copies into profiles/.  Every profiler run sits under a hard timeout (a hung rocprofv3 once cost 40 GPU-minutes).
set -u
TAG=${1:-r02}
STEPS=${2:-60}
COMMIT=${COMMIT:-unknown}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps $STEPS --warmup 10 --no-cpu-baseline --odom-frames 0 --no-vgicp --no-plane --no-traffic --no-fp64 --no-concurrent --sustain-s 0"
# the C4 (VGICP) and C2 (point-to-plane) legs of the bench on their own trace: their kernels' stats go to <tag>_c4_c2_kernel_stats.csv
BENCH_C4C2="python $ROOT/bench.py --steps 20 --warmup 10 --no-cpu-baseline --odom-frames 0 --no-traffic --no-fp64 --no-concurrent --sustain-s 0"
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/bench_trace.log" 2>&1
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- $BENCH > "$OUT/bench_pmc_fetch.log" 2>&1
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- $BENCH > "$OUT/bench_pmc_write.log" 2>&1
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c4c2" -o c4c2 -- $BENCH_C4C2 > "$OUT/bench_c4c2.log" 2>&1
cat > /tmp/odom_run.py <<PY
import sys; sys.path.insert(0, "$ROOT")
from small_gicp_amd import odometry
r = odometry.run_synthetic(12)
print({k: v for k, v in r.items() if k not in ("estimated", "ground_truth")})
PY
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/odom" -o odom -- python /tmp/odom_run.py > "$OUT/odom.log" 2>&1
cd "$ROOT"
python scripts/summarize_profile.py "$OUT" "$TAG" "$COMMIT" > "$OUT/${TAG}_rocprofv3_summary.txt" 2>&1
cat "$OUT/${TAG}_rocprofv3_summary.txt"
# the raw per-dispatch tables are large: keep the kernel stats + the condensed summaries only
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -name "*kernel_trace.csv" -delete
