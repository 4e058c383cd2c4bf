#!/usr/bin/env python3
"""Longest tile first on the hardware, with the TRUE costs: the same cold pass (every certificate ignored) three times at the same
pose — the first launch is in tile order, the following ones (SGA_LPT=2) in the order of the durations the previous one recorded.
Usage: SGA_LPT=2 python scripts/diag_lpt_hw.py [points]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import small_gicp_amd as sga

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
target, source, T_gt = sga.synthetic.registration_pair(n)
ctx = sga.default_context()
tgt, src = sga.PointCloud(target), sga.PointCloud(source)
sga.estimate_covariances(tgt, None, 20)
sga.estimate_covariances(src, None, 20)
tree = sga.KdTree(tgt)
st = sga.make_setting("GICP", max_correspondence_distance=1.0)
sga.set_warm_limit(-1.0)
pb = sga.Problem(tree, src)
half = np.eye(4)
half[:3, 3] = 0.5 * T_gt[:3, 3]
for name, T in (("identity", np.eye(4)), ("half way", half), ("ground truth", T_gt)):
    times = []
    for rep in range(4):
        ctx.set_profiling(1)
        r = pb.linearize(st.factor, T)
        times.append(ctx.kernel_ms()["search_ms"] * 1e3)
        ctx.set_profiling(0)
    print("%-12s search+factor kernel: %s us (first launch: tile order%s)" % (name, " ".join("%.1f" % t for t in times), "; then by the previous launch's durations" if os.environ.get("SGA_LPT") == "2" else ""), flush=True)
