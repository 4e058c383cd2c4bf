#!/bin/bash
# GPU box: per-stage wall vs kernel time of a C5 scan (VERDICT r5 #1a).  bash scripts/odom_stage_table.sh <tag>
# -> gpurun_out/<tag>_odom_stage_table.txt (copy into profiles/)
set -u
TAG=${1:-r06}
ROOT=$PWD
OUT=$ROOT/gpurun_out/stage_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o t -- python $ROOT/scripts/odom_stage_table.py ${FRAMES:-12} "$OUT/stages.json" > "$OUT/run.log" 2>&1
cd "$ROOT"
tail -20 "$OUT/run.log"
t=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)
python - "$t" "$OUT/stages.json" <<'PY' | tee gpurun_out/${TAG}_odom_stage_table.txt
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
st = json.load(open(sys.argv[2]))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
stages = [s for s in st["stages"] if s[1] != "align_iterations"]
iters = [s[2] for s in st["stages"] if s[1] == "align_iterations" and s[0] >= st["skip"]]
lo, hi = stages[0][2], stages[-1][3]
best = None
for name, off in (("monotonic", 0), ("boottime", st["boot_minus_mono"]), ("realtime", st["real_minus_mono"])):
    inside = sum(1 for a, b, _ in ks if lo + off <= a <= hi + off)
    if best is None or inside > best[0]:
        best = (inside, name, off)
inside, clock, off = best
print("# kernel timestamps are in the %s domain (%d of %d dispatches inside the staged region)" % (clock, inside, len(ks)))
agg = {}
import bisect
starts = [k[0] for k in ks]
for f, name, t0, t1 in stages:
    if f < st["skip"]:
        continue
    i0, i1 = bisect.bisect_left(starts, t0 + off), bisect.bisect_right(starts, t1 + off)
    kern = sum(ks[i][1] - ks[i][0] for i in range(i0, i1)) / 1e3
    span = (ks[i1 - 1][1] - ks[i0][0]) / 1e3 if i1 > i0 else 0.0
    a = agg.setdefault(name, {"wall": [], "kern": [], "n": [], "span": [], "names": {}})
    a["wall"].append((t1 - t0) / 1e3); a["kern"].append(kern); a["n"].append(i1 - i0); a["span"].append(span)
    for i in range(i0, i1):
        nm = ks[i][2].split("(")[0][-60:]
        d = a["names"].setdefault(nm, [0, 0.0]); d[0] += 1; d[1] += (ks[i][1] - ks[i][0]) / 1e3
mean = lambda v: sum(v) / max(1, len(v))
print("# C5 scan, per stage, each stage followed by a context synchronize; mean over %d frames; us" % len(agg["upload"]["wall"]))
print("%-16s %9s %9s %9s %9s" % ("stage", "wall", "kernels", "span", "launches"))
tw = tk = 0.0
for name, a in agg.items():
    print("%-16s %9.1f %9.1f %9.1f %9.1f" % (name, mean(a["wall"]), mean(a["kern"]), mean(a["span"]), mean(a["n"])))
    if name != "one_linearize":
        tw += mean(a["wall"]); tk += mean(a["kern"])
print("%-16s %9.1f %9.1f" % ("sum (no probe)", tw, tk))
print("align: %.2f LM iterations per scan" % mean(iters))
print("product chain (nothing synchronised between the stages): registration %.1f us, total %.1f us per scan" % (st["chain_reg_us"], st["chain_total_us"]))
for name, a in agg.items():
    nf = len(a["wall"])
    print("## %s" % name)
    for nm, (c, us) in sorted(a["names"].items(), key=lambda kv: -kv[1][1]):
        print("   %6.2f x %8.2f us  %s" % (c / nf, us / nf, nm))
PY
rm -rf "$OUT/trace"
