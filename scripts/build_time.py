import sys, time; sys.path.insert(0,".")
import numpy as np, small_gicp_amd as sga
ctx = sga.default_context()
for n in (1_000_000, 400_000, 100_000):
    t = sga.synthetic.scene(n, 1)
    c = sga.PointCloud(t)
    sga.estimate_covariances(c, None, 20)
    for rep in range(3):
        ms, tree = ctx.gpu_time_ms(lambda: sga.KdTree(c))
        t0 = time.perf_counter(); tr2 = sga.KdTree(c); ctx.synchronize(); wall = time.perf_counter() - t0
    ms_vg, d = ctx.gpu_time_ms(lambda: sga.voxelgrid_sampling(c, 0.25))
    src = sga.PointCloud(sga.synthetic.scene(n, 2))
    sga.estimate_covariances(src, None, 20)
    for rep in range(2):
        ms_pb, pb = ctx.gpu_time_ms(lambda: sga.Problem(tree, src))
    print("n=%d: kd build %.1f us GPU (%.1f us wall), voxel grid %.1f us, problem (source sort) %.1f us" % (n, 1e3*ms, 1e6*wall, 1e3*ms_vg, 1e3*ms_pb))
