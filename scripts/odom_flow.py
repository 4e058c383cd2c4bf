"""C5 as a flow of stages (PipelinedOdometry): ms/scan over (preprocessing workers) x (registration workers), poses against the sequential
driver's.  usage: python scripts/odom_flow.py [frames] [pinned|pageable] [PxR ...]"""
import sys

sys.path.insert(0, "/root/repo")
import numpy as np

from small_gicp_amd import api, odometry, synthetic

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
pinned = len(sys.argv) > 2 and sys.argv[2] == "pinned"
grid = [tuple(int(x) for x in a.split("x")) for a in sys.argv[3:]] or [(1, 1), (2, 1), (2, 2), (3, 2), (3, 3), (4, 3), (4, 4), (6, 4)]  # one process per entry: streams created earlier take hardware queues
scans = [synthetic.kitti_like_scan(f)[0] for f in range(frames)]
scans = [api.pinned_copy(s[:, :3], np.float32) if pinned else np.ascontiguousarray(s[:, :3], dtype=np.float32) for s in scans]
seq = odometry.OnlineOdometry()
ref = [seq.estimate(s) for s in scans]
print("sequential: registration %.3f total %.3f ms/scan" % (np.mean(seq.reg_ms[2:]), np.mean(seq.total_ms[2:])), flush=True)
for P, R in grid:
    od = odometry.PipelinedOdometry(workers=P, reg_workers=R)
    od.run(scans[:6])
    best = None
    for rep in range(3):
        poses, wall, iters = od.run(scans)
        same = all(np.array_equal(a, b) for a, b in zip(poses, ref))
        dmax = max(float(np.abs(a - b).max()) for a, b in zip(poses, ref))
        ms = 1e3 * wall / frames
        best = ms if best is None else min(best, ms)
    print("workers %d reg_workers %d: %.3f ms/scan (best of 3)  poses identical %s (max diff %.1e)  iterations %.2f" % (P, R, best, same, dmax, np.mean(iters)), flush=True)
    del od
