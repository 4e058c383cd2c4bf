#!/usr/bin/env python3
"""K1 search kernel A/B on one C3 registration: per-pass K1 time for the one-query-per-lane kernel and the queue-fed kernel at
several chunk sizes.  Usage: python scripts/diag_search.py [points]"""
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
ref = None
for mode in [(False, 0, 0)] + [(True, c, w) for c, w in ((1, 1), (2, 2), (4, 4), (8, 8), (2, 8), (4, 16), (16, 16))]:
    sga.set_search_mode(*mode)
    for rep in range(2):
        pb = sga.Problem(tree, src)
        rows = []
        prev = {"s": pb.pass_stats()}

        def lin(T):
            ctx.set_profiling(1)
            r = pb.linearize(st.factor, T)
            k = ctx.kernel_ms()
            s = pb.pass_stats()
            rows.append(("warm" if s["warm_passes"] > prev["s"]["warm_passes"] else "cold", k["search_ms"] * 1e3, k["linearize_ms"] * 1e3))
            prev["s"] = s
            ctx.set_profiling(0)
            return r

        res = sga.optimize(st, np.eye(4), lin, lambda T: pb.error(st.factor, T))
    corr, _ = pb.factors()
    if ref is None:
        ref = corr
    same = bool((corr == ref).all())
    print("mode %s: search us per pass: %s | avg search %.1f, avg K1 %.1f, iterations %d, same=%s" % (mode, " ".join("%s%.0f" % (r[0][0], r[1]) for r in rows), np.mean([r[1] for r in rows]), np.mean([r[2] for r in rows]), res.iterations, same), flush=True)
