import sys; sys.path.insert(0,".")
import numpy as np, small_gicp_amd as sga
ctx = sga.default_context()
for n in (1_000_000, 100_000):
    t = sga.synthetic.scene(n, 1)
    c = sga.PointCloud(t); tree = sga.KdTree(c)
    for k in (20, 10):
        for rep in range(3):
            ms, _ = ctx.gpu_time_ms(lambda: sga.estimate_covariances(c, tree, k))
        print("covariances k=%d n=%d: %.1f us" % (k, n, 1e3*ms))
