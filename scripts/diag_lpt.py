#!/usr/bin/env python3
"""Design study: would starting the expensive tiles first shorten a cold pass?  Per-tile work (max leaves scanned by any of its 64
lanes — a wave lasts as long as its longest lane) of the passes of one C3 registration, its correlation with the PREVIOUS pass's,
and a list-scheduling simulation on 8192 wave slots: launch order = tile order (today), = descending true cost (the bound), =
descending cost of the previous pass (what could be built).  Usage: python scripts/diag_lpt.py [points]"""
import heapq
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
sga.set_warm_limit(-1.0)  # every pass walks in full: the work of a cold pass at each pose
pb = sga.Problem(tree, src)
pb.search_stats(True)
costs = []


def lin(T):
    r = pb.linearize(st.factor, T)
    lv = pb.search_stats()
    m = len(lv) // 64 * 64
    costs.append(lv[:m].reshape(-1, 64).max(axis=1).astype(np.float64))
    return r


sga.optimize(st, np.eye(4), lin, lambda T: pb.error(st.factor, T))


def makespan(cost, order, slots=8192):
    """each wave occupies a slot for (2 + cost) units; waves start in `order` as slots free up"""
    h = [0.0] * slots
    heapq.heapify(h)
    end = 0.0
    for t in order:
        s = heapq.heappop(h)
        e = s + 2.0 + cost[t]
        end = max(end, e)
        heapq.heappush(h, e)
    return end


for k, c in enumerate(costs[:5]):
    tiles = np.arange(len(c))
    line = "pass %d: tile cost mean %.1f max %d | makespan in tile order %.1f, LPT by true cost %.1f (bound: mean load %.1f)" % (k, c.mean(), c.max(), makespan(c, tiles), makespan(c, np.argsort(-c, kind="stable")), (2.0 + c).sum() / 8192)
    if k > 0:
        p = costs[k - 1]
        line += " | corr with previous pass %.2f, LPT by previous pass %.1f" % (np.corrcoef(p, c)[0, 1], makespan(c, np.argsort(-p, kind="stable")))
    print(line, flush=True)
