#!/usr/bin/env python3
"""If every lane of a search wave owned Q queries and walked them back to back (a lane starts its next query when its current one is done,
independently of the other lanes), how full would the waves be?  Uses the leaves scanned per source point of the cold passes of one C3
registration (the measured proxy of a walk's length, scripts/diag_leaves.py) and two assignments of queries to lanes:
  block:   a wave owns 64 Q consecutive (spatially sorted) queries, lane l gets l, l + 64, l + 128, ...
  The wave's duration is the longest lane's sum; the lane utilisation is sum / (64 x that).
Usage: python scripts/diag_multi_query.py [points]"""
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
st = sga.make_setting("GICP", max_correspondence_distance=1.0, max_iterations=4, rotation_eps=0.0, translation_eps=0.0)
sga.set_search_mode(0)
sga.set_warm_limit(-1.0)  # every pass cold
pb = sga.Problem(tree, src)
pb.search_stats(True)
k = [0]


def lin(T):
    r = pb.linearize(st.factor, T)
    lv = pb.search_stats().astype(np.int64) + 1  # (+1: the fixed part of a walk — descent, result — in units of a leaf scan, roughly)
    out = []
    for Q in (1, 2, 4, 8):
        m = len(lv) // (64 * Q) * (64 * Q)
        t = lv[:m].reshape(-1, Q, 64).sum(axis=1)  # lane l of a wave: queries l, l + 64, ...
        wmax = t.max(axis=1)
        out.append("Q=%d util %.2f (time %.2f of Q=1)" % (Q, t.sum() / (64.0 * wmax.sum()), wmax.sum() / float(lv[:m].reshape(-1, 64).max(axis=1).sum())))
    print("pass %d: leaves/point mean %.2f | %s" % (k[0], lv.mean() - 1, " | ".join(out)), flush=True)
    k[0] += 1
    return r


sga.optimize(st, np.eye(4), lin, lambda T: pb.error(st.factor, T))
