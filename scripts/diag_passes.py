#!/usr/bin/env python3
"""Per-pass diagnostics of one C3 registration: motion since the previous linearization, kind of pass, K1 time (HIP events),
points re-searched.  Usage: python scripts/diag_passes.py [points] [warm_delta]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import small_gicp_amd as sga

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
if len(sys.argv) > 2:
    sga.set_warm_limit(float(sys.argv[2]))
target, source, T_gt = sga.synthetic.registration_pair(n)
ctx = sga.default_context()
tgt, src = sga.PointCloud(target), sga.PointCloud(source)
sga.estimate_covariances(tgt, None, 20)
sga.estimate_covariances(src, None, 20)
tree = sga.KdTree(tgt)
pb = sga.Problem(tree, src)
st = sga.make_setting(os.environ.get("DIAG_FACTOR", "GICP"), max_correspondence_distance=1.0, max_iterations=10, rotation_eps=0.0, translation_eps=0.0)
lo, hi = source.min(0).astype(np.float64), source.max(0).astype(np.float64)
corners = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])
for rep in range(2):
    prev = {"T": None, "stats": pb.pass_stats()}
    rows = []

    def lin(T):
        ctx.set_profiling(1)
        r = pb.linearize(st.factor, T)
        k = ctx.kernel_ms()
        s = pb.pass_stats()
        moved = None if prev["T"] is None else float(np.linalg.norm(corners @ (T[:3, :3] - prev["T"][:3, :3]).T + (T[:3, 3] - prev["T"][:3, 3]), axis=1).max())
        kind = "warm" if s["warm_passes"] > prev["stats"]["warm_passes"] else ("grid" if s["grid_passes"] > prev["stats"]["grid_passes"] else "cold")
        rows.append((moved, kind, k["linearize_ms"] * 1e3, k["search_ms"] * 1e3, s["walked_points"] - prev["stats"]["walked_points"], s["grid_open"] - prev["stats"]["grid_open"], s["grid_rings"] - prev["stats"]["grid_rings"], s["adj_queries"] - prev["stats"]["adj_queries"], s["adj_unsettled"] - prev["stats"]["adj_unsettled"]))
        prev["T"], prev["stats"] = T.copy(), s
        ctx.set_profiling(0)
        return r

    def err(T):
        return pb.error(st.factor, T)

    res = sga.optimize(st, np.eye(4), lin, err)
    for _ in range(2):
        lin(res.T_target_source)  # same pose again: (nearly) every certificate holds
    if rep == 1:
        for i, r in enumerate(rows):
            print("pass %d moved=%s %s K1=%.1fus search=%.1fus walked=%d grid_open=%d rings=%d adj=%d unsettled=%d" % (i, "-" if r[0] is None else "%.5f" % r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8]))
        print("grid cell %.4f m" % pb.pass_stats()["grid_cell_m"])
        print("total K1 %.1f us over %d passes -> avg %.1f us; limits %s" % (sum(r[2] for r in rows), len(rows), sum(r[2] for r in rows) / len(rows), sga.get_warm_limit()))
    pb2 = sga.Problem(tree, src)  # fresh state for the second repetition (warm-up effects only)
    pb = pb2
