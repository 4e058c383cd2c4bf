#!/bin/bash
# GPU box: how much does the whole-align rate of the policy (C3, default flags) vary, and with what — host topology, then policy_bench
# five times unbound and five times bound to the cores of NUMA node 0.
cd /root/repo
lscpu | egrep "Model name|Socket|NUMA|Thread|Core|^CPU\(s\)" ; ls /sys/devices/system/node | grep node; cat /sys/devices/system/node/node0/cpulist
python - <<'PY'
import json, os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, "scripts"); sys.path.insert(0, ".")
import policy_bench
import small_gicp_amd as sga
n = 1_000_000
target, source, _ = sga.synthetic.registration_pair(n)
tgt, src = sga.PointCloud(target), sga.PointCloud(source)
sga.estimate_covariances(tgt, None, 20); sga.estimate_covariances(src, None, 20)
clouds = (target, source, sga.api.sym6_from_mats(tgt.covs()), sga.api.sym6_from_mats(src.covs()))
all_cpus = sorted(os.sched_getaffinity(0))
node0 = []
for part in open("/sys/devices/system/node/node0/cpulist").read().strip().split(","):
    a, _, b = part.partition("-")
    node0 += list(range(int(a), int(b or a) + 1))
node0 = [c for c in node0 if c in all_cpus]
for label, cpus in (("unbound", all_cpus), ("node0", node0), ("unbound", all_cpus), ("node0", node0)):
    for rep in range(3):
        os.sched_setaffinity(0, cpus)
        r = policy_bench.run("GICP", n, reps=5, clouds=clouds)
        os.sched_setaffinity(0, all_cpus)
        print(label, len(cpus), "whole %.0f median %.0f policy_calls %.0f lean %.0f align_ms %s" % (r["whole_align_iterations_per_s"], r["whole_align_median_iterations_per_s"], r["policy_calls_iterations_per_s"], r["lean"]["whole_align_iterations_per_s"], r["align_ms"]), flush=True)
PY
