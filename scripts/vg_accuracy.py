"""error of the voxel-grid centroids (as stored: fp32) against the exactly rounded double mean, for the path selected by SGA_VG_HASH"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import small_gicp_amd as sga
from small_gicp_amd import synthetic
pts, _ = synthetic.kitti_like_scan(0)
p = pts[:, :3].astype(np.float32)
for leaf in (0.1, 0.25):
    d = sga.voxelgrid_sampling(sga.PointCloud(p), leaf).xyz()
    c = np.floor(p.astype(np.float64) * (1.0 / leaf)).astype(np.int64) + 2**20
    key = c[:, 0] | (c[:, 1] << 21) | (c[:, 2] << 42)
    order = np.argsort(key, kind="stable")
    ks, ps = key[order], p[order].astype(np.longdouble)
    starts = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
    sums = np.add.reduceat(ps, starts, axis=0)
    cnt = np.diff(np.r_[starts, len(ks)])[:, None]
    exact = (sums / cnt).astype(np.float64)
    want = exact.astype(np.float32)
    print("leaf %g: %d voxels, rows not equal to the correctly rounded mean: %d, max |stored - exact| / ulp: %.3f" % (leaf, len(d), (d != want).any(axis=1).sum(), (np.abs(d.astype(np.float64) - exact) / np.spacing(np.abs(want))).max()))
