#!/usr/bin/env python3
"""Adversarial inputs for the fast leaf scan (kd_search.hpp: 32-bit truncated keys + exact repeat of undecided queries): the
correspondences of a linearization must not depend on which scan ran.  Run once per setting of SGA_FAST_SCAN (read when the library
loads) with the same seed and compare the files:
    SGA_FAST_SCAN=1 python scripts/stress_fast_scan.py out1.npz; SGA_FAST_SCAN=0 python scripts/stress_fast_scan.py out0.npz
With --compare a b the script loads both and exits non-zero on any difference."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def datasets(rng):
    n = 120_000
    cube = rng.uniform(-20, 20, (n, 3))
    yield "uniform cube", cube, cube[rng.choice(n, 60_000)] + rng.normal(0, 0.05, (60_000, 3)), 1.0
    planes = np.concatenate([np.c_[rng.uniform(-30, 30, (n // 2, 2)), np.zeros(n // 2)], np.c_[rng.uniform(-30, 30, (n // 2, 2)), np.full(n // 2, 1e-4)]]) + rng.normal(0, 1e-7, (n, 3))
    yield "two planes 0.1 mm apart", planes, np.c_[rng.uniform(-30, 30, (50_000, 2)), rng.uniform(-0.3, 0.3, 50_000)], 1.0
    g = np.arange(-20, 21, dtype=np.float64)
    lattice = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    lattice = np.concatenate([lattice, lattice[rng.choice(len(lattice), 20_000)]])  # with duplicates
    yield "integer lattice with duplicates", lattice, np.concatenate([lattice[rng.choice(len(lattice), 20_000)] + 0.5, rng.uniform(-20, 20, (20_000, 3))]), 2.0
    far = rng.uniform(-5, 5, (n, 3)) + np.array([12000.0, -9000.0, 300.0])
    yield "far from the origin (fp32 resolution 1 mm)", far, far[rng.choice(n, 40_000)] + rng.normal(0, 0.01, (40_000, 3)), 1.0
    tiny = rng.uniform(-1e-3, 1e-3, (n, 3))
    yield "millimetre scale", tiny, tiny[rng.choice(n, 40_000)] + rng.normal(0, 1e-5, (40_000, 3)), -1.0
    blobs = np.concatenate([rng.normal(c, 0.02, (n // 40, 3)) for c in rng.uniform(-50, 50, (40, 3))] + [rng.uniform(-500, 500, (500, 3))])
    yield "dense blobs and far outliers, no rejector", blobs, np.concatenate([blobs[rng.choice(len(blobs), 30_000)] + rng.normal(0, 0.01, (30_000, 3)), rng.uniform(-500, 500, (5_000, 3))]), -1.0
    yield "63 points", cube[:63], cube[:200] + 0.01, 1.0
    yield "one point", cube[:1], cube[:70], -1.0


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--compare":
        a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
        bad = 0
        for k in a.files:
            if not np.array_equal(a[k], b[k]):
                bad += 1
                print("DIFFERENT:", k, int((a[k] != b[k]).sum()) if a[k].shape == b[k].shape else "shape")
        print("fast scan stress: %d arrays compared, %d different" % (len(a.files), bad))
        sys.exit(1 if bad else 0)
    import small_gicp_amd as sga

    rng = np.random.default_rng(1234)
    out = {}
    for name, tgt, src, maxd in datasets(rng):
        tree = sga.KdTree(sga.PointCloud(tgt.astype(np.float32)))
        pb = sga.Problem(tree, sga.PointCloud(src.astype(np.float32)))
        st = sga.make_setting("ICP", max_correspondence_distance=maxd)
        for k, T in enumerate((np.eye(4), np.array([[1, 0, 0, 0.013], [0, 1, 0, -0.004], [0, 0, 1, 0.002], [0, 0, 0, 1.0]]))):
            H, b, e, ninl = pb.linearize(st.factor, T)
            out["%s / pose %d / correspondences" % (name, k)] = pb.factors()[0]
            out["%s / pose %d / inliers" % (name, k)] = np.array([ninl])
        print("%-45s %7d target %6d source: %d inliers" % (name, len(tgt), len(src), ninl), flush=True)
    np.savez(sys.argv[1], **out)


if __name__ == "__main__":
    main()
