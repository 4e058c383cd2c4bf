import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np
from small_gicp_amd import odometry, synthetic, api
odom = odometry.OnlineOdometry()
for f in range(14):
    pts, _ = synthetic.kitti_like_scan(f)
    t0 = time.perf_counter()
    raw = api.PointCloud(pts, ctx=odom.ctx); odom.ctx.synchronize()
    t1 = time.perf_counter()
    del raw
    odom.estimate(pts)
    import ctypes as C
    st = (C.c_uint64 * 5)(); api.load().sga_allocator_stats(st)
    print("alloc stats malloc/stream/pool/pending/deferred", list(st), end="  ")
    print("frame %d upload %.3f ms  total %.3f reg %.3f" % (f, 1e3*(t1-t0), odom.total_ms[-1], odom.reg_ms[-1]))
