#!/usr/bin/env python3
"""Soak test of the exactness claim (DESIGN.md sections 3.1 - 3.3) at sizes and pose sequences the unit tests do not reach: random pose
chains shaped like LM runs on synthetic pairs of 70k ... 1M points; after EVERY linearization the correspondences of the product path
(warm passes, certificates, queue-fed kernel, fast leaf scan, longest-tile-first launches) must equal those of a problem that walks every
point from the root in every pass.  Usage: python scripts/soak_exactness.py [chains per size] [seed]"""
import os
import sys
import time

import numpy as np
from scipy.spatial.transform import Rotation

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import small_gicp_amd as sga

chains = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
default_limit = sga.get_warm_limit()
total_passes = total_diff = 0
t0 = time.time()
for n in (70_000, 300_000, 600_000, 1_000_000):
    target, source, T_gt = sga.synthetic.registration_pair(n)
    tgt, src = sga.PointCloud(target), sga.PointCloud(source)
    sga.estimate_covariances(tgt, None, 10)
    sga.estimate_covariances(src, None, 10)
    tree = sga.KdTree(tgt)
    for c in range(chains):
        kind = ("GICP", "ICP", "GICP", "GICP")[c % 4]
        st = sga.make_setting(kind, max_correspondence_distance=(1.0, 0.5, 2.0, 1.0)[c % 4])
        # a goal near the ground truth, approached with steps that shrink by a random factor per pass (large, medium, small, tiny, zero)
        goal = T_gt.copy()
        goal[:3, 3] += rng.normal(0, 0.05, 3)
        rv = Rotation.from_matrix(goal[:3, :3]).as_rotvec() + rng.normal(0, 0.003, 3)
        fr, f = [0.0], 0.0
        while len(fr) < 11:
            f += (1.0 - f) * rng.uniform(0.4, 0.97)
            fr.append(f)
        fr += [fr[-1], 1.0, 1.0]
        pw, pc = sga.Problem(tree, src), sga.Problem(tree, src)
        for k, f in enumerate(fr):
            T = np.eye(4)
            T[:3, :3] = Rotation.from_rotvec(rv * f).as_matrix()
            T[:3, 3] = goal[:3, 3] * f
            sga.set_warm_limit(default_limit)
            Hw, bw, ew, nw = pw.linearize(st.factor, T)
            cw = pw.factors()[0]
            sga.set_warm_limit(-1.0)
            Hc, bc, ec, nc = pc.linearize(st.factor, T)
            cc = pc.factors()[0]
            d = int((cw != cc).sum())
            total_passes += 1
            total_diff += d
            if d or nw != nc:
                print("MISMATCH n=%d chain %d pass %d: %d correspondences differ, inliers %d vs %d" % (n, c, k, d, nw, nc), flush=True)
        s = pw.pass_stats()
        print("n=%d chain %d (%s): %d passes, %d warm, %d points re-walked, all correspondences equal so far: %s" % (n, c, kind, len(fr), s["warm_passes"], s["walked_points"], total_diff == 0), flush=True)
sga.set_warm_limit(default_limit)
print("SOAK %s: %d passes, %d differing correspondences, %.0f s" % ("OK" if total_diff == 0 else "FAILED", total_passes, total_diff, time.time() - t0))
sys.exit(0 if total_diff == 0 else 1)
