#!/usr/bin/env python3
"""Leaves scanned per source point in every pass of one C3 registration (one-query-per-lane search kernel): how uneven are the
walks inside a wave of 64 neighbours?  Usage: python scripts/diag_leaves.py [points]"""
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
st = sga.make_setting("GICP", max_correspondence_distance=1.0, max_iterations=10, rotation_eps=0.0, translation_eps=0.0)
sga.set_search_mode(0)
pb = sga.Problem(tree, src)
pb.search_stats(True)
k = [0]


def lin(T):
    r = pb.linearize(st.factor, T)
    lv = pb.search_stats()
    m = len(lv) // 64 * 64
    t = lv[:m].reshape(-1, 64)
    wmax = t.max(axis=1)
    print("pass %d inliers %d: leaves/point mean %.2f p50 %d p90 %d p99 %d p99.9 %d max %d | walkers %.1f%% | per wave: mean of max %.1f, sum/(64*max) %.2f, waves with max>=16: %.1f%%, >=32: %.1f%%"
          % (k[0], r[3], lv.mean(), np.percentile(lv, 50), np.percentile(lv, 90), np.percentile(lv, 99), np.percentile(lv, 99.9), lv.max(), 100.0 * (lv > 0).mean(), wmax.mean(),
             t.sum() / max(1, (64 * wmax).sum()), 100.0 * (wmax >= 16).mean(), 100.0 * (wmax >= 32).mean()), flush=True)
    k[0] += 1
    return r


res = sga.optimize(st, np.eye(4), lin, lambda T: pb.error(st.factor, T))
