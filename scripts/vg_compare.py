"""voxel grid of KITTI-shaped scans: dumps the downsampled clouds (run once per SGA_VG_HASH setting), or compares two dumps"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        x, y = a[k], b[k]
        print(k, x.shape, y.shape, "max abs diff %.3e" % (np.abs(x - y).max() if x.shape == y.shape else float("nan")), "bit-equal rows %d of %d" % ((x == y).all(axis=1).sum() if x.shape == y.shape else -1, len(x)))
    sys.exit(0)
import small_gicp_amd as sga
from small_gicp_amd import synthetic
out = {}
for f in (0, 7):
    pts, _ = synthetic.kitti_like_scan(f)
    for leaf in (0.25, 0.1, 1.0):
        for rep in range(2):
            d = sga.voxelgrid_sampling(sga.PointCloud(pts[:, :3].astype(np.float32)), leaf).xyz()
        out["f%d_leaf%g" % (f, leaf)] = d
t, s, _ = synthetic.registration_pair(200_000)
out["c3_200k"] = sga.voxelgrid_sampling(sga.PointCloud(t + np.float32(1000.0)), 0.3).xyz()
np.savez(sys.argv[1], **out)
print("saved", {k: v.shape for k, v in out.items()})
