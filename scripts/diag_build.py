#!/usr/bin/env python3
"""Where the time of a first bind goes (C3 clouds): upload, covariance estimation, kd-tree + cell grid build, problem creation — first and repeated."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import small_gicp_amd as sga

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
target, source, T_gt = sga.synthetic.registration_pair(n)
ctx = sga.default_context()
for rep in range(3):
    t = [time.perf_counter()]
    tgt = sga.PointCloud(target); ctx.synchronize(); t.append(time.perf_counter())
    src = sga.PointCloud(source); ctx.synchronize(); t.append(time.perf_counter())
    sga.estimate_covariances(tgt, None, 20); ctx.synchronize(); t.append(time.perf_counter())
    sga.set_grid_mode(0)
    tree0 = sga.KdTree(tgt); ctx.synchronize(); t.append(time.perf_counter())
    sga.set_grid_mode(1)
    tree = sga.KdTree(tgt); ctx.synchronize(); t.append(time.perf_counter())
    pb = sga.Problem(tree, src); ctx.synchronize(); t.append(time.perf_counter())
    d = np.diff(t) * 1e3
    print("rep %d: upload target %.1f ms, source %.1f ms | covariances (temp tree + kNN) %.1f | kd-tree %.1f | kd-tree + cell grid %.1f | problem (sort by target leaf) %.1f" % (rep, *d))
    del pb, tree, tree0
