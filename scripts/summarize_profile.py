"""Condense rocprofv3 output (kernel stats CSV + PMC counter CSVs + the odometry trace) into a short text summary and the JSON files
bench.py / DESIGN.md cite:  summarize_profile.py <prof dir> <tag> <commit>"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

out, tag, commit = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "r02"), (sys.argv[3] if len(sys.argv) > 3 else "unknown")


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


def short(name):
    return name.split("(")[0][-70:]


summary = {"commit": commit}
for f in find("trace/**/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    shutil.copyfile(f, os.path.join(out, "%s_kernel_stats.csv" % tag))
    print("== kernel stats of the bench workload (C3):", os.path.relpath(f, out))
    for r in rows[:12]:
        print("%-70s calls=%s avg_ns=%s total_ns=%s pct=%s" % (short(r.get("Name", "")), r.get("Calls"), r.get("AverageNs"), r.get("TotalDurationNs"), r.get("Percentage")))
        summary.setdefault("kernel_stats", []).append({"name": r.get("Name"), "calls": int(r.get("Calls", 0)), "avg_ns": float(r.get("AverageNs", 0)), "pct": float(r.get("Percentage", 0))})

# per-dispatch durations of the K1 kernels from the trace: cold / warm split of the search kernel by duration order is not possible
# here (the host knows the kind); report the distribution instead
for f in find("trace/**/*kernel_trace.csv"):
    durs = defaultdict(list)
    for r in csv.DictReader(open(f)):
        kn = r.get("Kernel_Name", "")
        for key in ("search_linearize_kernel", "certify_linearize_kernel", "nn_search_queue_kernel", "nn_search_kernel", "linearize_kernel", "error_kernel", "reduce_rows_kernel"):
            if key in kn:
                durs[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
                break
    print("== per-dispatch durations (us): min / median / mean / max")
    for k, v in durs.items():
        v.sort()
        print("%-22s n=%d  %.1f / %.1f / %.1f / %.1f" % (k, len(v), v[0], v[len(v) // 2], sum(v) / len(v), v[-1]))
        summary.setdefault("durations_us", {})[k] = {"n": len(v), "min": v[0], "median": v[len(v) // 2], "mean": sum(v) / len(v), "max": v[-1]}

K1_SEARCH = ("search_linearize_kernel", "certify_linearize_kernel", "nn_search_queue_kernel", "nn_search_kernel")  # one launch of these per linearization pass


def k1_kind(name):
    for k in K1_SEARCH:
        if k in name:
            return k
    return "linearize_kernel" if "linearize_kernel" in name else None


totals = {}
for ctr, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    agg = defaultdict(lambda: [0.0, 0])
    for f in find("pmc_%s/**/*counter_collection.csv" % ctr):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != cname:
                continue
            k = k1_kind(r.get("Kernel_Name", ""))
            if k is None:
                continue
            agg[k][0] += float(r.get("Counter_Value", 0))
            agg[k][1] += 1
    print("== PMC", cname, "(KB per launch, raw)")
    for k, (v, c) in sorted(agg.items()):
        print("%-28s per_launch=%.1f (n=%d)" % (k, v / max(c, 1), c))
        summary.setdefault("pmc", []).append({"kernel": k, "counter": cname, "per_launch": v / max(c, 1), "launches": c})
    totals[cname] = agg

if totals.get("FETCH_SIZE") and totals.get("WRITE_SIZE"):
    passes = sum(c for k, (v, c) in totals["FETCH_SIZE"].items() if k in K1_SEARCH)
    fetch_kb = sum(v for v, c in totals["FETCH_SIZE"].values())
    passes_w = sum(c for k, (v, c) in totals["WRITE_SIZE"].items() if k in K1_SEARCH)
    write_kb = sum(v for v, c in totals["WRITE_SIZE"].values())
    if passes > 0 and passes_w > 0:
        hbm = int((2.0 * fetch_kb / passes + write_kb / passes_w) * 1024)
        traffic = {
            "kernel": "K1 = the launch(es) of one linearization pass, config C3 1M<->1M: sga::search_linearize_kernel<float, GICP> (cold passes and the first warm ones: search + factors in one launch), "
                      "sga::nn_search_queue_kernel<float, warm, GICP> (warm passes after motions of 2 - 20 mm: certificate check, queue-fed walks, factors), sga::certify_linearize_kernel<float, GICP> (warm passes "
                      "after smaller motions: certificate check inside the streaming factor kernel), and on the non-fused paths sga::nn_search_kernel + sga::linearize_kernel; averaged over the passes of whole registrations",
            "source": "scripts/profile_gpu.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py",
            "commit": commit,
            "passes": passes,
            "FETCH_SIZE_KB_per_launch_raw": {k: v / max(c, 1) for k, (v, c) in totals["FETCH_SIZE"].items()},
            "WRITE_SIZE_KB_per_launch_raw": {k: v / max(c, 1) for k, (v, c) in totals["WRITE_SIZE"].items()},
            "correction": "MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE tallies 128-B requests at 64 B -> x2 on the read side; WRITE_SIZE as reported",
            "hbm_bytes_per_launch": hbm,
            "algorithmic_bytes_per_launch": 100000000,
        }
        json.dump(traffic, open(os.path.join(out, "%s_k1_traffic.json" % tag), "w"), indent=1)
        print("== K1 HBM traffic per pass: %.1f MB (algorithmic 100 MB) -> %.2fx" % (hbm / 1e6, hbm / 1e8))

for f in find("c4c2/**/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    shutil.copyfile(f, os.path.join(out, "%s_c4_c2_kernel_stats.csv" % tag))
    print("== kernel stats of a bench run WITH the C4 (VGICP, linearize_kernel<float, GICP, voxel map>) and C2 (PLANE_ICP) legs:", os.path.relpath(f, out))
    for r in rows[:14]:
        print("%-70s calls=%s avg_ns=%s total_ns=%s pct=%s" % (short(r.get("Name", "")), r.get("Calls"), r.get("AverageNs"), r.get("TotalDurationNs"), r.get("Percentage")))
        summary.setdefault("c4_c2_kernel_stats", []).append({"name": r.get("Name"), "calls": int(r.get("Calls", 0)), "avg_ns": float(r.get("AverageNs", 0)), "pct": float(r.get("Percentage", 0))})

for f in find("odom/**/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    shutil.copyfile(f, os.path.join(out, "%s_odom_kernel_stats.csv" % tag))
    frames = 12.0
    calls = sum(int(r["Calls"]) for r in rows) / frames
    total = sum(float(r["TotalDurationNs"]) for r in rows) / 1e3 / frames
    print("== odometry leg (C5, 12 scans): %.1f launches / scan, %.1f us of kernels / scan" % (calls, total))
    top = [{"name": short(r["Name"]), "calls_per_scan": int(r["Calls"]) / frames, "us_per_scan": float(r["TotalDurationNs"]) / 1e3 / frames} for r in rows[:12]]
    for t in top:
        print("%8.1f us/scan %5.1f calls/scan  %s" % (t["us_per_scan"], t["calls_per_scan"], t["name"]))
    json.dump({"commit": commit, "frames": frames, "launches_per_scan": calls, "kernel_us_per_scan": total, "top_kernels": top, "source": "scripts/profile_gpu.sh: rocprofv3 --kernel-trace --stats of small_gicp_amd.odometry.run_synthetic(12)"},
              open(os.path.join(out, "%s_odom_trace.json" % tag), "w"), indent=1)
json.dump(summary, open(os.path.join(out, "%s_rocprofv3_summary.json" % tag), "w"), indent=1)
