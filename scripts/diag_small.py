"""cold / warm pass times (HIP events) of the GICP registration on small source clouds against the 1M-point target (shards) and of C2-size problems"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import small_gicp_amd as sga
ctx = sga.default_context()
def run(tgt_n, src_n, factor="GICP"):
    target, source, T_gt = sga.synthetic.registration_pair(tgt_n)
    tgt, src = sga.PointCloud(target), sga.PointCloud(source[:src_n] if src_n < tgt_n else source)
    if factor == "GICP":
        sga.estimate_covariances(tgt, None, 20); sga.estimate_covariances(src, None, 20)
    else:
        sga.estimate_normals(tgt, None, 20)
    tree = sga.KdTree(tgt)
    pb = sga.Problem(tree, src)
    st = sga.make_setting(factor, max_correspondence_distance=1.0, max_iterations=10, rotation_eps=0.0, translation_eps=0.0)
    for _ in range(3): pb.align(st, np.eye(4))
    ctx.set_profiling(1)
    for _ in range(5): pb.align(st, np.eye(4))
    k = ctx.kernel_ms(); ctx.set_profiling(0)
    print("%s target %d source %d: K1 %.1f us (cold %.1f x%d, warm %.1f x%d)" % (factor, tgt_n, src.size(), 1e3*k["linearize_ms"], 1e3*k["cold_ms"], k["cold_calls"], 1e3*k["warm_ms"], k["warm_calls"]), flush=True)
run(100_000, 100_000, "PLANE_ICP")
run(100_000, 100_000, "GICP")
run(1_000_000, 125_000)
run(1_000_000, 250_000)
