#!/usr/bin/env python3
"""What ONE rank of an N-way sharded C3 registration does per pass, measured on one GPU: the 1M-point target index whole, a contiguous
1/N slice of the (Morton-ordered) 1M-point source, the poses of the unsharded registration.  K1 (HIP events) per pass for N = 1, 2, 4, 8:
the numbers DESIGN.md section 6 builds its scaling model from.  Usage: diag_shards.py [points]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import small_gicp_amd as sga

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
target, source, T_gt = sga.synthetic.registration_pair(n)
# the bench's order for sharded runs: Morton order of the source
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import morton_order  # noqa: E402

source = source[morton_order(source)]
ctx = sga.default_context()
tgt, src = sga.PointCloud(target), sga.PointCloud(source)
sga.estimate_covariances(tgt, None, 20)
sga.estimate_covariances(src, None, 20)
tree = sga.KdTree(tgt)
st = sga.make_setting("GICP", max_correspondence_distance=1.0, max_iterations=10, rotation_eps=0.0, translation_eps=0.0)
# poses of the unsharded registration
poses = []
full = sga.Problem(tree, src)


def lin_rec(T):
    poses.append(T.copy())
    return full.linearize(st.factor, T)


sga.optimize(st, np.eye(4), lin_rec, lambda T: full.error(st.factor, T))
for N in (1, 2, 4, 8):
    rows = []
    for r in (0, N // 2):  # two of the shards: the first and a middle one
        lo, hi = r * n // N, (r + 1) * n // N
        pb = sga.Problem(tree, src.slice(lo, hi - lo))
        for rep in range(2):
            pb = sga.Problem(tree, src.slice(lo, hi - lo))
            times = []
            for T in poses:
                ctx.set_profiling(1)
                pb.linearize(st.factor, T)
                times.append(ctx.kernel_ms()["linearize_ms"] * 1e3)
                ctx.set_profiling(0)
        rows.append(times)
    t = np.array(rows)
    print("N=%d shard=%d points: per-pass K1 us (shard 0): %s | mean over passes: shard0 %.1f, middle shard %.1f" % (N, n // N, " ".join("%.0f" % x for x in t[0]), t[0].mean(), t[1].mean()))
