#!/bin/bash
# per-kernel time of one command under rocprofv3 (run on the GPU box): kstats.sh <tag> <command...>
TAG=$1; shift
ROOT=$PWD
OUT=$ROOT/gpurun_out/kstats_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- "$@" > "$OUT/log.txt" 2>&1
cd "$ROOT"
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(__import__("os").environ.get("KSTATS_TOP", "14"))]:
    print("%-90s calls %5s avg %9.1f us  total %9.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
find "$OUT" -name "*kernel_trace.csv" -delete
