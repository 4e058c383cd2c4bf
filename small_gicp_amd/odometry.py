"""Scan-to-scan GICP odometry (config C5), the protocol of the reference's benchmark
(src/benchmark/odometry_benchmark_small_gicp_omp.cpp:16-49 + include/small_gicp/benchmark/benchmark_odom.hpp:49-82):

  per frame:  voxelgrid_sampling(0.25 m)                      -> "total" time only (benchmark_odom.hpp:60-66)
              index build + estimate_covariances(k = 20)       -> registration time
              Registration<GICPFactor>::align(prev, cur, prev_tree, Identity); T_world = T_world * T     -> registration time
              the current scan (with its index and covariances) becomes the next target: each scan is preprocessed once.

Everything runs on the GPU through the C-ABI; only the raw scan crosses PCIe.
"""
import os
import time

import numpy as np

from . import api


class OnlineOdometry:
    """shard = (rank, world): multi-GPU form (BASELINE config C5).  Every rank preprocesses the whole scan (voxel grid, index,
    covariances: replicated), registers only its contiguous shard of the source against the replicated target
    and the context's RCCL communicator (Context.comm_init) sums the shards' systems once per linearization / error pass, so every rank
    computes the same pose."""

    def __init__(self, downsampling_resolution=0.25, num_neighbors=20, max_correspondence_distance=1.0, ctx=None, shard=None):
        self.shard = shard
        self.res = downsampling_resolution
        self.k = num_neighbors
        self.setting = api.make_setting("GICP", max_correspondence_distance=max_correspondence_distance)
        self.max_dist = max_correspondence_distance
        # every step of a scan runs on ONE context in stream-ordered mode: no host wait between the index build, the covariances and the
        # registration.  The mode changes what "returned" means for every user of that context, so the estimator takes a context of its own
        # unless it is handed one (the sharded bench leg: the context that carries the communicator), whose mode it restores in close().
        self._own_ctx = ctx is None
        self.ctx = ctx or api.Context(0)
        self._prev_mode = self.ctx.set_stream_ordered(True)
        self.target = None  # (cloud, tree)
        self.T_world = np.eye(4)
        self.reg_ms = []
        self.total_ms = []
        self.iterations = []

    def close(self):
        """Give a borrowed context its previous mode back (synchronises it)."""
        if self.ctx is not None and not self._own_ctx:
            self.ctx.set_stream_ordered(self._prev_mode)
        self.target = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def estimate(self, points):
        """points: (N,3|4) float32 in the sensor frame. Returns T_world_sensor of this scan."""
        t0 = time.perf_counter()
        raw = api.PointCloud(points, ctx=self.ctx)
        cloud = api.voxelgrid_sampling(raw, self.res)
        self.ctx.synchronize()
        t1 = time.perf_counter()
        tree = api.KdTree(cloud)
        api.estimate_covariances(cloud, tree, self.k)
        if self.target is not None:
            tgt_cloud, tgt_tree = self.target
            src = tree  # the scan by its own index: its kd order is spatially coherent, the problem takes it as it is (no sort)
            if self.shard is not None:
                rank, world = self.shard
                n = cloud.size()
                lo, hi = rank * n // world, (rank + 1) * n // world
                src = cloud.slice(lo, hi - lo)
            res = api.Problem(tgt_tree, src, np.eye(4)).align(self.setting, np.eye(4))
            self.T_world = self.T_world @ res.T_target_source
            self.iterations.append(res.iterations + 1)
        self.ctx.synchronize()
        t2 = time.perf_counter()
        self.target = (cloud, tree)
        self.reg_ms.append(1e3 * (t2 - t1))
        self.total_ms.append(1e3 * (t2 - t0))
        return self.T_world.copy()


class ModelOdometry:
    """Scan-to-MODEL VGICP odometry (src/benchmark/odometry_benchmark_small_vgicp_model_omp.cpp:12-57): the target is one
    GaussianVoxelMap accumulating every registered scan (incremental insert with the estimated pose + LRU removal of voxels the
    sensor has left behind); each new scan is registered against it starting from the previous pose, then inserted."""

    def __init__(self, downsampling_resolution=0.25, num_neighbors=20, voxel_resolution=1.0, max_correspondence_distance=1.0, ctx=None, model="gaussian"):
        """model = "gaussian": GaussianVoxelMap / VGICP (odometry_benchmark_small_vgicp_model_omp.cpp); "flat": IncrementalVoxelMap<
        FlatContainerCov> / GICP against the stored points (odometry_benchmark_small_gicp_model_omp.cpp)."""
        self.model = model
        self.res = downsampling_resolution
        self.k = num_neighbors
        self.voxel_resolution = voxel_resolution
        self.setting = api.make_setting("GICP", max_correspondence_distance=max_correspondence_distance)
        self.ctx = ctx or api.default_context()
        self.voxelmap = None
        self.T_world = np.eye(4)
        self.reg_ms, self.total_ms, self.iterations = [], [], []

    def estimate(self, points):
        t0 = time.perf_counter()
        raw = api.PointCloud(points, ctx=self.ctx)
        cloud = api.voxelgrid_sampling(raw, self.res)
        self.ctx.synchronize()
        t1 = time.perf_counter()
        api.estimate_covariances(cloud, None, self.k)
        if self.voxelmap is None:  # the very first frame
            self.voxelmap = (api.IncrementalVoxelMapCov if self.model == "flat" else api.GaussianVoxelMap)(self.voxel_resolution, ctx=self.ctx)
            self.voxelmap.insert(cloud)
        else:
            res = api.Problem(self.voxelmap, cloud, self.T_world).align(self.setting, self.T_world)
            self.T_world = res.T_target_source
            self.iterations.append(res.iterations + 1)
            self.voxelmap.insert(cloud, self.T_world)
            self.ctx.synchronize()
            self.reg_ms.append(1e3 * (time.perf_counter() - t1))
        self.ctx.synchronize()
        self.total_ms.append(1e3 * (time.perf_counter() - t0))
        return self.T_world.copy()


def run_synthetic_model(num_frames=20, **kw):
    """ModelOdometry over the frozen synthetic sequence; absolute trajectory error against the generator's ground truth."""
    from . import synthetic

    odom = ModelOdometry(**kw)
    est, gt = [], []
    T0 = None
    for f in range(num_frames):
        pts, Tws = synthetic.kitti_like_scan(f)
        if T0 is None:
            T0 = Tws
        est.append(odom.estimate(pts))
        gt.append(np.linalg.inv(T0) @ Tws)
    ate = [float(np.linalg.norm(e[:3, 3] - g[:3, 3])) for e, g in zip(est, gt)]
    skip = 2 if num_frames > 4 else 0
    return {
        "frames": num_frames,
        "registration_ms_per_scan": float(np.mean(odom.reg_ms[skip:])) if len(odom.reg_ms) > skip else float("nan"),
        "total_ms_per_scan": float(np.mean(odom.total_ms[skip + 1 :])) if len(odom.total_ms) > skip + 1 else float("nan"),
        "total_ms_per_scan_median": float(np.median(odom.total_ms[skip + 1 :])) if len(odom.total_ms) > skip + 1 else float("nan"),
        "mean_iterations": float(np.mean(odom.iterations)) if odom.iterations else 0.0,
        "ate_trans_m_max": max(ate),
        "num_voxels": odom.voxelmap.size(),
        "estimated": est,
        "ground_truth": gt,
    }


class PipelinedOdometry:
    """The same scan-to-scan odometry as a flow of stages over HIP streams (one context each) — the HIP-stream analogue of the reference's
    TBB flow graph (src/benchmark/odometry_benchmark_small_gicp_tbb_flow.cpp:55-141):

      preprocess_node   (`workers` threads, frames i, i + workers, ...: upload, voxel grid, index build, covariances; :61-67)
      pairing_node      (frame i-1 is the target of frame i; :73-78)
      registration_node (`reg_workers` threads: pair i is registered from the identity whatever pair i-1 gave, so the pairs are independent
                         and the reference runs this node with unlimited concurrency too; :81-97)
      output_node       (the relative poses multiplied up in frame order; :103-110)

    A frame is a chain of ~30 dependent launches that leaves the GPU mostly idle, so several chains interleave almost for free.  Poses
    are identical to OnlineOdometry's (same kernels, same order of operations per frame and per pair; the products in frame order)."""

    def __init__(self, downsampling_resolution=0.25, num_neighbors=20, max_correspondence_distance=1.0, device=0, workers=2, depth=None, reg_workers=2):
        # defaults: 2 x 2 workers — the measured optimum on one MI355X (profiles/r06_flow_cpp_grid.txt); more streams make every kernel slower
        self.res = downsampling_resolution
        self.k = num_neighbors
        self.setting = api.make_setting("GICP", max_correspondence_distance=max_correspondence_distance)
        self.ctx_pre = [api.Context(device) for _ in range(max(1, workers))]
        for c in self.ctx_pre:  # a frame's chain is enqueued without host waits; whoever consumes its cloud / index waits for their events (common.hpp: Ready)
            c.set_stream_ordered(True)
        self.ctx_reg = [api.Context(device) for _ in range(max(1, reg_workers))]
        self.depth = max(depth or 6, len(self.ctx_pre) + len(self.ctx_reg) + 1)

    def _preprocess(self, points, ctx):
        raw = api.PointCloud(points, ctx=ctx)
        cloud = api.voxelgrid_sampling(raw, self.res)
        tree = api.KdTree(cloud)
        api.estimate_covariances(cloud, tree, self.k)
        return cloud, tree

    def run(self, scans):
        """scans: sequence of (N,3|4) float32 arrays.  Returns (poses T_world_sensor, wall seconds, iterations per registration)."""
        import threading

        scans = list(scans)
        n = len(scans)
        P, R = len(self.ctx_pre), len(self.ctx_reg)
        ready = {}  # frame -> (cloud, tree), until both of its pairs are registered
        rel = {}    # frame -> (T_(frame-1) frame, iterations)
        cv = threading.Condition()
        state = {"next_pair": 1, "error": None}  # every pair below next_pair is registered (the producers run at most `depth` frames ahead of it)

        def fail(ex):
            with cv:
                if state["error"] is None:
                    state["error"] = ex
                cv.notify_all()

        def producer(w):
            try:
                for i in range(w, n, P):
                    with cv:
                        cv.wait_for(lambda: i < state["next_pair"] + self.depth or state["error"] is not None)
                        if state["error"] is not None:
                            return
                    item = self._preprocess(scans[i], self.ctx_pre[w])
                    with cv:
                        ready[i] = item
                        cv.notify_all()
            except BaseException as ex:  # noqa: BLE001
                fail(ex)

        def registrar(r):
            try:
                ctx = self.ctx_reg[r]
                for i in range(1 + r, n, R):
                    with cv:
                        cv.wait_for(lambda: (i in ready and i - 1 in ready) or state["error"] is not None)
                        if state["error"] is not None:
                            return
                        tgt, src = ready[i - 1], ready[i]
                    # the registration runs on its own context / stream, which waits for the events behind the producers' work
                    res = api.Problem(tgt[1], src[1], np.eye(4), ctx=ctx).align(self.setting, np.eye(4))
                    del tgt, src
                    with cv:
                        rel[i] = (res.T_target_source, res.iterations + 1)
                        while state["next_pair"] in rel:  # pairs up to here are done: their frames below the last one have served as source and target
                            ready.pop(state["next_pair"] - 1, None)
                            state["next_pair"] += 1
                        if state["next_pair"] == n:
                            ready.pop(n - 1, None)
                        cv.notify_all()
            except BaseException as ex:  # noqa: BLE001
                fail(ex)

        # The threads spend their time inside ctypes calls (GIL released) and need the GIL for microseconds in between; with CPython's default
        # switch interval (5 ms) a thread coming back from a call can wait that long for one that is running bytecode.  SGA_PIPE_SWITCH_S
        # (default 2e-5 s) for the duration of the run.
        import os
        import sys

        old_switch = sys.getswitchinterval()
        sys.setswitchinterval(float(os.environ.get("SGA_PIPE_SWITCH_S", "2e-5")))
        t0 = time.perf_counter()
        threads = [threading.Thread(target=producer, args=(w,), daemon=True) for w in range(P)]
        threads += [threading.Thread(target=registrar, args=(r,), daemon=True) for r in range(R)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        sys.setswitchinterval(old_switch)
        if state["error"] is not None:
            raise state["error"]
        for c in self.ctx_reg:
            c.synchronize()
        # output node: the relative poses multiplied up in frame order (the same products as OnlineOdometry's)
        poses, iters = [], []
        T_world = np.eye(4)
        for i in range(n):
            if i > 0:
                T_world = T_world @ rel[i][0]
                iters.append(rel[i][1])
            poses.append(T_world.copy())
        wall = time.perf_counter() - t0
        return poses, wall, iters


def run_synthetic(num_frames=20, pinned=False, **kw):
    """Drive OnlineOdometry over the frozen KITTI-shaped synthetic sequence (small_gicp_amd.synthetic.kitti_like_scan).
    pinned: the scans are held in pinned host memory (api.pinned_copy) before the timed loop, like a driver that reads its scans into
    memory from sga_host_alloc — the reference's benchmark holds them in host memory too (benchmark/benchmark_odom.hpp:36-47); the upload
    then has no CPU pass."""
    from . import synthetic

    odom = OnlineOdometry(**kw)
    est, gt, sizes = [], [], []
    # every scan is in host memory before the loop starts, as in the reference's driver (benchmark/benchmark_odom.hpp:36-47; the generator
    # is ~50 ms of host work per scan: run between the frames it would leave the GPU idle and clocked down)
    scans, T0 = [], None
    for f in range(num_frames):
        pts, Tws = synthetic.kitti_like_scan(f)
        scans.append(api.pinned_copy(pts[:, :3], np.float32) if pinned else np.ascontiguousarray(pts[:, :3], dtype=np.float32))
        if T0 is None:
            T0 = Tws
        sizes.append(len(pts))
        gt.append(np.linalg.inv(T0) @ Tws)
    for pts in scans:
        est.append(odom.estimate(pts))
    # relative pose error per frame pair (what scan-to-scan registration controls)
    rpe_t, rpe_r = [], []
    for i in range(1, num_frames):
        de = np.linalg.inv(est[i - 1]) @ est[i]
        dg = np.linalg.inv(gt[i - 1]) @ gt[i]
        E = np.linalg.inv(dg) @ de
        rpe_t.append(float(np.linalg.norm(E[:3, 3])))
        rpe_r.append(float(np.arccos(min(1.0, max(-1.0, (np.trace(E[:3, :3]) - 1) / 2)))))
    skip = 2 if num_frames > 4 else 1  # first frames carry first-touch allocations
    return {
        "frames": num_frames,
        "scans_in_pinned_host_memory": bool(pinned),
        "points_per_scan": float(np.mean(sizes)),
        "registration_ms_per_scan": float(np.mean(odom.reg_ms[skip:])) if len(odom.reg_ms) > skip else float("nan"),
        "total_ms_per_scan": float(np.mean(odom.total_ms[skip:])) if len(odom.total_ms) > skip else float("nan"),
        "mean_iterations": float(np.mean(odom.iterations)) if odom.iterations else 0.0,
        "rpe_trans_m_mean": float(np.mean(rpe_t)) if rpe_t else 0.0,
        "rpe_rot_rad_mean": float(np.mean(rpe_r)) if rpe_r else 0.0,
        "estimated": est,
        "ground_truth": gt,
    }


def run_synthetic_pairs(num_frames, rank, world, device=0, **kw):
    """Frame-pair parallelism over ranks (the other way to spread C5 over GPUs): under the reference's protocol every pair (scan f-1, scan f)
    is registered from the identity (odometry_benchmark_small_gicp_omp.cpp:16-49), so the pairs are independent — rank r takes the
    contiguous block of frames [r F / W, (r + 1) F / W), preprocesses those scans plus the one before the block, and registers its pairs
    on a context of its own with no collective at all.  Returns this rank's wall time, its relative poses {f: T_(f-1) f} and frame count;
    the job's ms/scan is max over ranks of the wall time / F.  flow=(P, R): the rank runs its block as a flow of stages (PipelinedOdometry: P
    preprocessing and R registration workers with a stream each) instead of frame by frame."""
    import time as _time

    from . import synthetic

    lo, hi = rank * num_frames // world, (rank + 1) * num_frames // world
    flow = kw.pop("flow", None)
    # generate the scans first: the generator is host work that a real stream would not pay
    scans = {f: synthetic.kitti_like_scan(f)[0] for f in range(max(lo - 1, 0), hi)}
    if flow:  # (preprocessing workers, registration workers): the rank's block through the flow form (PipelinedOdometry) on its device
        frames = sorted(scans)
        pipe = PipelinedOdometry(device=device, workers=flow[0], reg_workers=flow[1], **kw)
        pipe.run([scans[f] for f in frames[: min(4, len(frames))]])  # first-touch allocations, code objects
        poses, el, _ = pipe.run([scans[f] for f in frames])
        rel = {f: np.linalg.inv(poses[k - 1]) @ poses[k] for k, f in enumerate(frames) if k > 0 and f >= max(lo, 1)}
        return {"seconds": el, "frames": hi - lo, "relative_poses": rel}
    odom = OnlineOdometry(ctx=api.Context(device), **kw)
    rel = {}
    odom.ctx.synchronize()
    t0 = _time.perf_counter()
    prev_T = None
    for f in sorted(scans):
        T = odom.estimate(scans[f])
        if prev_T is not None and f >= max(lo, 1):
            rel[f] = np.linalg.inv(prev_T) @ T
        prev_T = T
    odom.ctx.synchronize()
    el = _time.perf_counter() - t0
    odom.close()
    return {"seconds": el, "frames": hi - lo, "relative_poses": rel}


def run_synthetic_pipelined(num_frames=20, pinned=False, **kw):
    """Throughput of the two-stream pipeline on the synthetic sequence: wall time / frame with all frames in flight."""
    from . import synthetic

    scans = [synthetic.kitti_like_scan(f)[0] for f in range(num_frames)]
    if pinned:
        scans = [api.pinned_copy(s[:, :3], np.float32) for s in scans]
    odom = PipelinedOdometry(**kw)
    odom.run(scans[: min(3, num_frames)])  # first-touch allocations, code objects
    poses, wall, iters = odom.run(scans)
    return {"frames": num_frames, "ms_per_scan": 1e3 * wall / num_frames, "estimated": poses, "mean_iterations": float(np.mean(iters)) if iters else 0.0}


def _cpp_driver(source, workdir, num_frames):
    """Write the synthetic sequence as KITTI .bin files under workdir/velodyne (once) and compile examples/<source> against the in-tree
    library.  Returns (executable, dataset directory)."""
    import os
    import subprocess

    from . import _lib, synthetic

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    data = os.path.join(workdir, "velodyne")
    os.makedirs(data, exist_ok=True)
    for f in range(num_frames):
        name = os.path.join(data, "%06d.bin" % f)
        if os.path.exists(name):
            continue
        pts, _ = synthetic.kitti_like_scan(f)
        v = np.zeros((len(pts), 4), "<f4")
        v[:, :3] = pts[:, :3]
        v.tofile(name)
    exe = os.path.join(workdir, os.path.splitext(source)[0])
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", source), "-o", exe, "-L" + libdir, "-lsmall_gicp_amd",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"])
    return exe, data


def _read_trajectory(path):
    poses = []
    for ln in open(path):
        T = np.eye(4)
        T[:3, :4] = np.array(ln.split(), dtype=np.float64).reshape(3, 4)
        poses.append(T)
    return poses


def run_synthetic_cpp(num_frames=20, workdir=None, downsampling_resolution=0.25, num_neighbors=20):
    """The same sequence through the C++ driver examples/odometry_benchmark.cpp (the reference's benchmark protocol over
    include/small_gicp_amd.hpp): writes the scans as KITTI .bin files, compiles the driver with g++ against the in-tree library, runs it and
    parses its report and trajectory.  Returns registration / total ms per scan as the driver measured them and the poses."""
    import re
    import shutil
    import subprocess
    import tempfile

    own = workdir is None
    workdir = workdir or tempfile.mkdtemp(prefix="sga_odom_cpp_")
    try:
        exe, data = _cpp_driver("odometry_benchmark.cpp", workdir, num_frames)
        traj = os.path.join(workdir, "traj.txt")
        p = subprocess.run([exe, data, traj, "--num_neighbors", str(num_neighbors), "--downsampling_resolution", str(downsampling_resolution), "--max_frames", str(num_frames)], capture_output=True, text=True, timeout=600)
        if p.returncode != 0:
            raise RuntimeError("odometry_benchmark failed: " + p.stdout[-1000:] + p.stderr[-1000:])
        m = re.search(r"registration_time_stats=([0-9.eE+-]+) \+- ([0-9.eE+-]+) \[msec/scan\]\s+total_throughput=([0-9.eE+-]+) \+- ([0-9.eE+-]+) \[msec/scan\]\s+mean_iterations=([0-9.eE+-]+)", p.stdout)
        if m is None:
            raise RuntimeError("no report in the driver's output: " + p.stdout[-1000:])
        return {"frames": num_frames, "registration_ms_per_scan": float(m.group(1)), "registration_ms_std": float(m.group(2)), "total_ms_per_scan": float(m.group(3)), "mean_iterations": float(m.group(5)),
                "estimated": _read_trajectory(traj), "driver": "examples/odometry_benchmark.cpp (C++ over include/small_gicp_amd.hpp), all scans read into host memory first"}
    finally:
        if own:
            shutil.rmtree(workdir, ignore_errors=True)


def run_synthetic_cpp_flow(num_frames=20, workdir=None, downsampling_resolution=0.25, num_neighbors=20, preprocess_workers=2, registration_workers=2, pinned=False, repeat=3, env=None):
    """The same sequence through the C++ FLOW driver examples/odometry_benchmark_flow.cpp — the throughput protocol of the reference's TBB
    flow-graph engine (odometry_benchmark_small_gicp_tbb_flow.cpp:50-141): preprocessing and registration stages on threads with a HIP
    stream each, pairs registered side by side, poses multiplied up in frame order.  Returns the best total_throughput [ms/scan] of the runs
    after the first (which carries code-object loads and first allocations), every run's figure and the trajectory of the last run."""
    import re
    import shutil
    import subprocess
    import tempfile

    own = workdir is None
    workdir = workdir or tempfile.mkdtemp(prefix="sga_odom_flow_")
    try:
        exe, data = _cpp_driver("odometry_benchmark_flow.cpp", workdir, num_frames)
        traj = os.path.join(workdir, "traj_flow.txt")
        cmd = [exe, data, traj, "--num_neighbors", str(num_neighbors), "--downsampling_resolution", str(downsampling_resolution), "--max_frames", str(num_frames), "--preprocess_workers", str(preprocess_workers),
               "--registration_workers", str(registration_workers), "--repeat", str(max(2, repeat))] + (["--pinned"] if pinned else [])
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, **(env or {})))
        if p.returncode != 0:
            raise RuntimeError("odometry_benchmark_flow failed: " + p.stdout[-1000:] + p.stderr[-1000:])
        runs = [(float(a), float(b), float(c)) for a, b, c in re.findall(r"total_throughput=([0-9.eE+-]+) \[msec/scan\]\s+frame_latency=([0-9.eE+-]+) \[msec\]\s+mean_iterations=([0-9.eE+-]+)", p.stdout)]
        if len(runs) < 2:
            raise RuntimeError("no report in the driver's output: " + p.stdout[-1000:])
        best = min(runs[1:])
        return {"frames": num_frames, "ms_per_scan": best[0], "frame_latency_ms": best[1], "mean_iterations": best[2], "runs_ms_per_scan": [r[0] for r in runs], "preprocess_workers": preprocess_workers,
                "registration_workers": registration_workers, "scans_in_pinned_host_memory": bool(pinned), "estimated": _read_trajectory(traj),
                "driver": "examples/odometry_benchmark_flow.cpp (C++ threads over include/small_gicp_amd.hpp, a HIP stream per worker), all scans read into host memory first; best of the runs after the first"}
    finally:
        if own:
            shutil.rmtree(workdir, ignore_errors=True)


class Summarizer:
    """mean +- std (last=...) of a stream, formatted like benchmark.hpp:36-79."""

    def __init__(self):
        self.n, self.sum, self.sq, self.last = 0, 0.0, 0.0, 0.0

    def push(self, x):
        self.n += 1
        self.sum += x
        self.sq += x * x
        self.last = x

    def __str__(self):
        if self.n == 0:
            return "nan +- nan (last=nan)"
        mean = self.sum / self.n
        var = max(0.0, (self.sq - mean * self.sum) / self.n)
        return "%.3f +- %.3f (last=%.3f)" % (mean, var ** 0.5, self.last)


def run_kitti(dataset_path, output_path=None, num_frames=None, downsampling_resolution=0.25, num_neighbors=20, quiet=False):
    """The reference's odometry benchmark on a directory of KITTI `.bin` scans (src/benchmark/odometry_benchmark.cpp:20-97 with the
    small_gicp engine, odometry_benchmark_small_gicp_omp.cpp:16-57): prints the same parameter / report lines and writes the
    trajectory in the KITTI text format.  Returns the list of poses T_world_sensor."""
    from . import io

    names = io.list_kitti_scans(dataset_path, num_frames or 1000000)
    if not quiet:
        print("dataset_path=%s" % dataset_path)
        print("registration_engine=small_gicp_amd")
        print("num_neighbors=%d" % num_neighbors)
        print("downsampling_resolution=%g" % downsampling_resolution)
        print("num_frames=%d" % len(names))
    odom = OnlineOdometry(downsampling_resolution=downsampling_resolution, num_neighbors=num_neighbors)
    reg, tot = Summarizer(), Summarizer()
    traj = []
    for i, name in enumerate(names):
        if i and i % 256 == 0 and not quiet:
            print("registration_time_stats=%s [msec/scan]  total_throughput=%s [msec/scan]" % (reg, tot))
        pts = io.read_points(name)
        traj.append(odom.estimate(pts[:, :3]))
        reg.push(odom.reg_ms[-1])
        tot.push(odom.total_ms[-1])
    if not quiet:
        print("done!")
        print("registration_time_stats=%s [msec/scan]  total_throughput=%s [msec/scan]" % (reg, tot))
    if output_path:
        io.write_trajectory(output_path, traj)
    return traj


def main(argv=None):
    import argparse

    ap = argparse.ArgumentParser(description="scan-to-scan GICP odometry on the GPU: odometry <dataset_path> <output_path> [options]")
    ap.add_argument("dataset_path", help="directory of KITTI .bin scans, or 'synthetic' for the frozen synthetic sequence")
    ap.add_argument("output_path", nargs="?", default=None, help="trajectory file (KITTI text format)")
    ap.add_argument("--num_frames", type=int, default=None)
    ap.add_argument("--num_neighbors", type=int, default=20)
    ap.add_argument("--downsampling_resolution", type=float, default=0.25)
    a = ap.parse_args(argv)
    if a.dataset_path == "synthetic":
        from . import io

        r = run_synthetic(a.num_frames or 20, downsampling_resolution=a.downsampling_resolution, num_neighbors=a.num_neighbors)
        print("registration_time=%.3f [msec/scan]  total=%.3f [msec/scan]  rpe_trans=%.4f m  rpe_rot=%.5f rad" % (r["registration_ms_per_scan"], r["total_ms_per_scan"], r["rpe_trans_m_mean"], r["rpe_rot_rad_mean"]))
        if a.output_path:
            io.write_trajectory(a.output_path, r["estimated"])
    else:
        run_kitti(a.dataset_path, a.output_path, a.num_frames, a.downsampling_resolution, a.num_neighbors)


if __name__ == "__main__":
    main()
