import sys; sys.path.insert(0, "/root/repo")
import numpy as np, small_gicp_amd as sga
ctx = sga.default_context()
for n in (1_000_000, 100_000, 250_000, 125_000):
    t = sga.synthetic.scene(n, 1)
    tree = sga.KdTree(sga.PointCloud(t)); ctx.synchronize()
    print("scene n=%d spacing %.6f" % (n, tree.spacing()))
    for s in (0.01, 10.0):
        tr = sga.KdTree(sga.PointCloud((t.astype(np.float64) * s).astype(np.float32))); ctx.synchronize()
        print("   scaled x%g: %.6f  ratio %.6f" % (s, tr.spacing(), tr.spacing() / tree.spacing() / s))
d = np.load("/root/repo/tests/golden/c1_points.npz")
c, tr = sga.preprocess_points(d["target"], 0.25, 10); ctx.synchronize(); print("C1 downsampled spacing", tr.spacing())
scan = sga.synthetic.kitti_like_scan(3)[0]
c, tr = sga.preprocess_points(scan, 0.25, 20); ctx.synchronize(); print("C5 scan spacing", tr.spacing(), c.size())
