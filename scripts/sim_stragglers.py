#!/usr/bin/env python3
"""Design study: how many loop trips would a cold pass save if a wave handed its last walking lanes (<= Y of 64) to waves made of such
stragglers only?  Input: the leaves scanned per source point in each pass of a C3 registration (every pass forced cold).  A wave's trips
~ the most leaves any of its ACTIVE lanes still has to scan; stragglers are regrouped 64 at a time in tile order and, per level, handed
on again.  Output per pass: wave-trips now, with one level, with levels until dry, and the lower bound (all lanes always busy).
Usage: python scripts/sim_stragglers.py [points]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import small_gicp_amd as sga

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
target, source, T_gt = sga.synthetic.registration_pair(n)
tgt, src = sga.PointCloud(target), sga.PointCloud(source)
sga.estimate_covariances(tgt, None, 20)
sga.estimate_covariances(src, None, 20)
tree = sga.KdTree(tgt)
st = sga.make_setting("GICP", max_correspondence_distance=1.0, max_iterations=10, rotation_eps=0.0, translation_eps=0.0)
sga.set_search_mode(0)
sga.set_warm_limit(-1.0)
pb = sga.Problem(tree, src)
pb.search_stats(True)
passes = []


def lin(T):
    r = pb.linearize(st.factor, T)
    passes.append(pb.search_stats().astype(np.int64))
    return r


sga.optimize(st, np.eye(4), lin, lambda T: pb.error(st.factor, T))


def level(counts, Y):
    """counts: remaining leaves per lane, grouped 64 per wave (padded with 0).  Returns (trips of these waves, remaining of the lanes handed on)."""
    m = (len(counts) + 63) // 64 * 64
    c = np.zeros(m, np.int64)
    c[: len(counts)] = counts
    c = c.reshape(-1, 64)
    if Y <= 0:
        return int(c.max(axis=1).sum()), np.zeros(0, np.int64)
    srt = np.sort(c, axis=1)[:, ::-1]
    stop = srt[:, Y]  # trips until at most Y lanes are left: the (Y+1)-th largest count
    rest = c - stop[:, None]
    return int(stop.sum()), rest[rest > 0]


for k, lv in enumerate(passes[:4]):
    now, _ = level(lv, 0)
    ideal = lv.sum() / 64.0
    line = "pass %d: leaf scans per query %.2f | wave-trips now %d (lanes busy %.0f %%), lower bound %d" % (k, lv.mean(), now, 100.0 * ideal / now, ideal)
    for Y in (8, 16, 24):
        t0, rest = level(lv, Y)
        t1, _ = level(rest, 0)
        total, r, levels = t0, rest, 0
        while len(r):
            t, r = level(r, Y if len(r) > 64 * 64 else 0)
            total += t
            levels += 1
        line += " | Y=%d: one level %d (%.0f %%), %d levels %d (%.0f %%), %d stragglers" % (Y, t0 + t1, 100.0 * (t0 + t1) / now, levels, total, 100.0 * total / now, len(rest))
    print(line, flush=True)
