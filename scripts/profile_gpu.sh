#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: kernel-trace stats + separate PMC passes for the bench workload.
# Outputs land in gpurun_out/prof_<tag>/ ; copy the summaries you want judged into profiles/.
set -u
TAG=${1:-r01}
STEPS=${2:-60}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $OLDPWD/bench.py --steps $STEPS --warmup 10 --no-cpu-baseline --odom-frames 0"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/bench_trace.log" 2>&1
# PMC: FETCH_SIZE (3 TCC slots) and WRITE_SIZE (2) in separate passes, kernel-trace only (no sys/hip/hsa tracing with --pmc)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- $BENCH > "$OUT/bench_pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- $BENCH > "$OUT/bench_pmc_write.log" 2>&1
cd "$OLDPWD"
python scripts/summarize_profile.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# the raw per-dispatch tables are large: keep the kernel stats + the condensed summaries only
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -name "*kernel_trace.csv" -delete
