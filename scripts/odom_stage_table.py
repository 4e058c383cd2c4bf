"""Per-stage wall time vs kernel time of ONE scan of the C5 odometry chain (VERDICT r5 #1a).

Run under `rocprofv3 --kernel-trace` (scripts/odom_stage_table.sh does) so that the kernels of every stage can be summed from the
trace: each stage is bracketed with a pair of marker launches (`sga_debug_marker` does not exist — the stage boundaries are recovered
from the host timestamps written next to the trace instead: the trace's timestamps and time.clock_gettime_ns(CLOCK_MONOTONIC /
BOOTTIME) share the clock domain rocprofv3 reports in).  Without a trace the script still prints the wall table.

Stages (src/benchmark/odometry_benchmark_small_gicp_omp.cpp:16-49): upload, voxel grid 0.25 m, kd-tree, covariances k = 20,
problem creation, every LM iteration of align().  Each stage is followed by a context synchronize, so "wall" is the stage's own
latency with nothing overlapped; the unsynchronised chain of the product (OnlineOdometry.estimate) is timed beside it.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402

from small_gicp_amd import api, odometry, synthetic  # noqa: E402


def now_ns():
    return time.clock_gettime_ns(time.CLOCK_MONOTONIC)


BOOT_MINUS_MONO = time.clock_gettime_ns(time.CLOCK_BOOTTIME) - time.clock_gettime_ns(time.CLOCK_MONOTONIC)
REAL_MINUS_MONO = time.clock_gettime_ns(time.CLOCK_REALTIME) - time.clock_gettime_ns(time.CLOCK_MONOTONIC)


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    out = sys.argv[2] if len(sys.argv) > 2 else None
    scans = [synthetic.kitti_like_scan(f)[0] for f in range(frames)]
    ctx = api.Context(0)
    ctx.set_stream_ordered(True)
    setting = api.make_setting("GICP", max_correspondence_distance=1.0)
    lib = api.load()
    stages = []  # (frame, name, t0_ns, t1_ns)

    def stage(f, name, fn):
        t0 = now_ns()
        r = fn()
        ctx.synchronize()
        t1 = now_ns()
        stages.append((f, name, t0, t1))
        return r

    prev = None
    for f, pts in enumerate(scans):
        raw = stage(f, "upload", lambda: api.PointCloud(pts, ctx=ctx))
        cloud = stage(f, "voxelgrid", lambda: api.voxelgrid_sampling(raw, 0.25))
        tree = stage(f, "kdtree", lambda: api.KdTree(cloud))
        stage(f, "covariances", lambda: api.estimate_covariances(cloud, tree, 20))
        if prev is not None:
            pb = stage(f, "problem", lambda: api.Problem(prev[1], tree, np.eye(4)))
            # the LM loop of align(), one stage per outer iteration: optimizer.hpp:83-149 through the C-ABI's own optimizer is one call,
            # so the iterations are split by timing whole align() and dividing by its iteration count beside a linearize-only probe
            res = stage(f, "align", lambda: pb.align(setting, np.eye(4)))
            stages.append((f, "align_iterations", res.iterations + 1, 0))
            T = res.T_target_source
            stage(f, "one_linearize", lambda: pb.linearize(setting.factor, T))
        prev = (cloud, tree)
    # the product's own chain, nothing synchronised in between
    odom = odometry.OnlineOdometry(ctx=api.Context(0))
    for pts in scans:
        odom.estimate(pts)
    skip = 3
    table = {}
    for f, name, t0, t1 in stages:
        if f < skip:
            continue
        if name == "align_iterations":
            table.setdefault(name, []).append(t0)
        else:
            table.setdefault(name, []).append((t1 - t0) / 1e3)
    wall = {k: float(np.mean(v)) for k, v in table.items()}
    print("stage wall (us, mean over frames %d..%d, each stage synchronised):" % (skip, frames - 1))
    for k, v in wall.items():
        print("  %-18s %8.1f" % (k, v))
    print("product chain: registration %.1f us, total %.1f us per scan" % (1e3 * np.mean(odom.reg_ms[skip:]), 1e3 * np.mean(odom.total_ms[skip:])))
    if out:
        json.dump({"boot_minus_mono": BOOT_MINUS_MONO, "real_minus_mono": REAL_MINUS_MONO, "stages": stages, "wall_us": wall, "skip": skip, "frames": frames, "chain_reg_us": 1e3 * float(np.mean(odom.reg_ms[skip:])), "chain_total_us": 1e3 * float(np.mean(odom.total_ms[skip:]))}, open(out, "w"))


if __name__ == "__main__":
    main()
