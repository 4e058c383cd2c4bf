#!/usr/bin/env python3
"""bench.py — registration-iterations/s of the GICP hot path on MI355X (BASELINE.json metric, config C3).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (config.workload = "C3"): GICP with per-point covariances (k = 20), 1M target points <-> 1M source points PER GPU,
synthetic planar scene (small_gicp_amd/synthetic.py, frozen in SURVEY.md §8d), max correspondence distance 1.0 m.
One STEP = one outer Levenberg-Marquardt iteration of Registration<GICPFactor>::align (registration/optimizer.hpp:100-144 of
the reference): 1x linearize (transform + exact NN + per-pair H/b/e + reduction) + the LM trial(s) (1x error pass each, normally
one) + the 6x6 solve on the host.  Registrations are restarted from the identity every 10 steps so the search workload stays
the real one.  Inputs (clouds, covariances, search index) are resident in HBM before the timed region.

Multi-GPU (weak scaling): the target + index are replicated, every rank owns an independent 1M-point source shard of an
N x 1M-point source cloud, the 30-double accumulator (21 H, 6 b, e, inliers) is all-reduced with RCCL once per linearize and
one double per error pass; every rank runs the same host LM on the reduced numbers.  value = N x steps / time, i.e. 1M-point
registration iterations per second summed over ranks ("iters_per_sec_job" is the plain iteration rate of the N x 1M job).

Extra objects on the JSON line: "roofline" (K1, algorithmic bytes / HIP-event time vs 8 TB/s) and "cpu_baseline" (the CPU oracle
— a restatement of the reference's OpenMP path, oracle/ — timed on this box's host cores on the same clouds; rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_POINT = {"linearize_gicp": 100, "error_gicp": 52}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PROFILE_PERIOD = 7  # coprime with ITERS_PER_ALIGN: the sampled passes cover every iteration index of a registration
ITERS_PER_ALIGN = 10


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--points", type=int, default=1_000_000, help="points per cloud per GPU (C3 = 1M)")
    ap.add_argument("--neighbors", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=10, help="outer LM iterations of the CPU baseline sample")
    ap.add_argument("--math", default="fp32", choices=["fp32", "fp64"])
    ap.add_argument("--odom-frames", type=int, default=12, help="frames of the KITTI-shaped scan-to-scan odometry leg (config C5); 0 = skip")
    ap.add_argument("--no-vgicp", action="store_true", help="skip the VGICP (config C4) leg")
    ap.add_argument("--force-dist", action="store_true", help="use the torch.distributed/RCCL path even at world size 1 (validation)")
    return ap.parse_args()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch  # must precede loading the HIP library: one HIP runtime per process
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    import small_gicp_amd as sga

    n = args.points
    # ---- data: identical target on every rank, an independent source resample per rank ----
    T_gt = sga.synthetic.gt_transform()
    target = sga.synthetic.scene(n, 1)
    src_world = sga.synthetic.scene(n, 2 + rank).astype(np.float64)
    Ti = np.linalg.inv(T_gt)
    source = (src_world @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)

    if use_dist:
        stream = torch.cuda.current_stream().cuda_stream
        ctx = sga.Context(local_rank, stream=stream)
    else:
        ctx = sga.Context(0)

    # ---- preprocessing on the GPU (untimed): covariances k = 20, search index, spatially sorted source ----
    t0 = time.perf_counter()
    tgt = sga.PointCloud(target, ctx=ctx)
    src = sga.PointCloud(source, ctx=ctx)
    sga.estimate_covariances(tgt, None, args.neighbors)
    sga.estimate_covariances(src, None, args.neighbors)
    tree = sga.KdTree(tgt)
    problem = sga.Problem(tree, src, np.eye(4))
    ctx.synchronize()
    prep_s = time.perf_counter() - t0

    setting = sga.make_setting("GICP", max_correspondence_distance=1.0, max_iterations=ITERS_PER_ALIGN, rotation_eps=0.0, translation_eps=0.0, math_mode=args.math)

    native_comm = False
    if use_dist:
        # native path: the library all-reduces its accumulators with RCCL on its own stream (no Python in the iteration loop);
        # torch.distributed only carries the 128-byte communicator id, the barriers and the max-over-ranks of the wall time
        try:
            ids = [sga.Context.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            ctx.comm_init(world, rank, ids[0])
            native_comm = True
        except Exception as ex:  # noqa: BLE001
            print("bench: native RCCL communicator unavailable (%r); falling back to torch.distributed callbacks" % (ex,), file=sys.stderr)

    if use_dist and not native_comm:
        acc = torch.zeros(sga._lib.ACCUM_DOUBLES, dtype=torch.float64, device="cuda")
        acc1 = torch.zeros(1, dtype=torch.float64, device="cuda")

        def lin_cb(T):
            problem.linearize_async(setting.factor, T, acc.data_ptr())
            dist.all_reduce(acc)
            return sga.unpack_accumulator(acc.cpu().numpy())

        def err_cb(T):
            problem.error_async(setting.factor, T, acc1.data_ptr())
            dist.all_reduce(acc1)
            return float(acc1.cpu()[0])

        def run_align(max_iters):
            s = sga.make_setting("GICP", max_correspondence_distance=1.0, max_iterations=max_iters, rotation_eps=0.0, translation_eps=0.0, math_mode=args.math)
            return sga.optimize(s, np.eye(4), lin_cb, err_cb)

    else:

        def run_align(max_iters):
            s = sga.make_setting("GICP", max_correspondence_distance=1.0, max_iterations=max_iters, rotation_eps=0.0, translation_eps=0.0, math_mode=args.math)
            return problem.align(s, np.eye(4))

    def run_steps(k):
        done = 0
        last = None
        while done < k:
            last = run_align(min(ITERS_PER_ALIGN, k - done))
            done += last.iterations + 1
        return done, last

    def barrier():
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()
        ctx.synchronize()

    run_steps(args.warmup)
    ctx.set_profiling(PROFILE_PERIOD)  # HIP events around every PROFILE_PERIOD-th pass, inside the timed region
    barrier()
    t0 = time.perf_counter()
    steps_done, last = run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    kms = ctx.kernel_ms()
    ctx.set_profiling(False)
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.cpu()[0])

    # sanity: the pose the timed registrations converge to
    E = np.linalg.inv(last.T_target_source) @ T_gt
    pose_err_t = float(np.linalg.norm(E[:3, 3]))
    pose_err_r = float(np.arccos(min(1.0, max(-1.0, (np.trace(E[:3, :3]) - 1) / 2))))

    if rank == 0:
        iters_per_sec_job = steps_done / elapsed
        value = iters_per_sec_job * world
        lin_us = kms["linearize_ms"] * 1e3
        err_us = kms["error_ms"] * 1e3
        achieved = (ALG_BYTES_PER_POINT["linearize_gicp"] * n) / (lin_us * 1e-6) / 1e9 if lin_us > 0 else None
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "k1_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:  # noqa: BLE001
                traffic = None
        out = {
            "metric": "GICP iterations/sec (1M<->1M pts)",
            "value": value,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": steps_done,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / steps_done,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.math == "fp32" else "f64",
            "data": "synthetic",
            "config": {
                "workload": "C3: GICP, per-point covariances k=20, %d target <-> %d source points per GPU, max_corr_dist 1.0 m" % (n, n),
                "step": "1 outer LM iteration = linearize + error pass(es) + host 6x6 solve; restart from identity every %d steps" % ITERS_PER_ALIGN,
                "parallelism": ("source sharded x%d, target index replicated, RCCL all-reduce of 30 doubles per linearize + 1 per error pass (%s)" % (world, "native ncclAllReduce on the library stream" if native_comm else "torch.distributed callbacks")) if use_dist else "single GPU",
                "source_points_total": n * world,
            },
            "iters_per_sec_job": iters_per_sec_job,
            "roofline": {
                "kernel": "K1 = nn_search_kernel<float> + linearize_kernel<float, GICP> (the two back-to-back launches of one linearize pass)",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                "traffic": traffic,
                "alg_bytes_per_launch": ALG_BYTES_PER_POINT["linearize_gicp"] * n,
                "avg_launch_us": lin_us,
                "launches_timed": kms["linearize_calls"],
                "nn_search_kernel_avg_us": kms["search_ms"] * 1e3,
                "linearize_kernel_avg_us": (kms["linearize_ms"] - kms["search_ms"]) * 1e3,
                "cold_pass_avg_us": kms["cold_ms"] * 1e3,
                "cold_passes_timed": kms["cold_calls"],
                "warm_pass_avg_us": kms["warm_ms"] * 1e3,
                "warm_passes_timed": kms["warm_calls"],
                "pass_stats": problem.pass_stats(),
                "error_kernel_avg_us": err_us,
                "error_kernel_achieved_GBs": (ALG_BYTES_PER_POINT["error_gicp"] * n) / (err_us * 1e-6) / 1e9 if err_us > 0 else None,
            },
            "preprocess_s": prep_s,
            "final_pose_error": {"trans_m": pose_err_t, "rot_rad": pose_err_r},
        }
        if world == 1 and not use_dist and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sga, tgt, src, n, args)
            if out["cpu_baseline"] and out["cpu_baseline"].get("value"):
                out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        if world == 1 and not use_dist and not args.no_vgicp:
            out["vgicp_c4"] = vgicp_leg(sga, ctx, tgt, src, args)
        if world == 1 and not use_dist and args.odom_frames > 1:
            out["kitti_odom"] = odometry_leg(sga, args)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def vgicp_leg(sga, ctx, tgt, src, args):
    """Config C4: VGICP against a GaussianVoxelMap (0.5 m voxels) of the same target, same step definition as the headline; the
    voxel lookup replaces the tree search, so a pass is the factor kernel alone.  Extra information, not the headline metric."""
    try:
        t0 = time.perf_counter()
        vm = sga.GaussianVoxelMap(0.5, ctx=ctx)
        vm.insert(tgt)
        problem = sga.Problem(vm, src, np.eye(4))
        ctx.synchronize()
        build_s = time.perf_counter() - t0
        s = sga.make_setting("GICP", max_correspondence_distance=1.0, max_iterations=ITERS_PER_ALIGN, rotation_eps=0.0, translation_eps=0.0, math_mode=args.math)
        for _ in range(2):
            problem.align(s, np.eye(4))
        ctx.set_profiling(PROFILE_PERIOD)
        ctx.synchronize()
        t0 = time.perf_counter()
        steps = 0
        while steps < 100:
            steps += problem.align(s, np.eye(4)).iterations + 1
        ctx.synchronize()
        el = time.perf_counter() - t0
        kms = ctx.kernel_ms()
        ctx.set_profiling(False)
        return {"value": steps / el, "unit": "iterations/s", "ms_per_step": 1e3 * el / steps, "voxelmap_build_s": build_s, "linearize_kernel_avg_us": kms["linearize_ms"] * 1e3, "error_kernel_avg_us": kms["error_ms"] * 1e3,
                "workload": "C4: VGICP, GaussianVoxelMap(0.5 m) of the 1M-point target, 1M source points"}
    except Exception as ex:  # noqa: BLE001
        return {"error": repr(ex)}


def odometry_leg(sga, args):
    """Second half of the BASELINE metric: ms/scan of scan-to-scan GICP odometry on the KITTI-shaped synthetic stream (C5), GPU vs
    the CPU oracle following the same protocol (downsample 0.25 m -> covariances k = 20 -> GICP against the previous scan)."""
    try:
        from small_gicp_amd import odometry

        r = odometry.run_synthetic(args.odom_frames)
        out = {k: v for k, v in r.items() if k not in ("estimated", "ground_truth")}
        out["unit"] = "ms/scan"
        try:
            pr = odometry.run_synthetic_pipelined(max(args.odom_frames, 36))
            same = all(np.abs(a - b).max() < 1e-9 for a, b in zip(pr["estimated"], r["estimated"]))
            out["pipelined_total_ms_per_scan"] = pr["ms_per_scan"]  # throughput with 3 preprocessing streams + 1 registration stream in flight
            out["pipelined_poses_identical"] = bool(same)
        except Exception as ex:  # noqa: BLE001
            out["pipelined_error"] = repr(ex)
        out["protocol"] = "src/benchmark/odometry_benchmark_small_gicp_omp.cpp:16-49: registration = index build + covariances + align; total adds the 0.25 m voxel grid"
        if not args.no_cpu_baseline:
            from oracle import orc

            ncpu = os.cpu_count() or 1
            threads = max(1, min(32, ncpu))
            prev = None
            reg_ms = []
            for f in range(min(4, args.odom_frames)):
                pts, _ = sga.synthetic.kitti_like_scan(f)
                down = orc.voxelgrid_sampling(pts, 0.25)
                t0 = time.perf_counter()
                cloud = orc.Cloud(down, tree=True)
                cloud.estimate_normals_covariances(20, threads)
                if prev is not None:
                    s = orc.default_setting(factor_kind=orc.GICP, num_threads=threads)
                    orc.align(prev, cloud, s)
                reg_ms.append(1e3 * (time.perf_counter() - t0))
                prev = cloud
            out["cpu_registration_ms_per_scan"] = float(np.mean(reg_ms[1:])) if len(reg_ms) > 1 else None
            out["cpu_threads"] = threads
        return out
    except Exception as ex:  # noqa: BLE001
        return {"error": repr(ex)}


def cpu_baseline(sga, tgt, src, n, args):
    """CPU baseline on the SAME clouds and covariances, all host threads, timed region = the optimizer loop only (index build
    excluded, as on the GPU side).  kind "reference": the unmodified reference code (registration_helper.cpp align ->
    Registration<GICPFactor, ParallelReductionOMP>) from oracle/_ref, when that library travelled with the repository;
    kind "port": the oracle's restatement of the same path (oracle/), otherwise."""
    try:
        from oracle import orc, ref

        orc.build()
        tp = tgt.xyz().astype(np.float64)
        sp = src.xyz().astype(np.float64)
        tcov = tgt.covs()[:, :3, :3]
        scov = src.covs()[:, :3, :3]
        ncpu = os.cpu_count() or 1
        counts = sorted({ncpu, max(1, ncpu // 2), max(1, ncpu // 4), max(1, ncpu // 8)}, reverse=True)
        use_ref = ref.available()
        t0 = time.perf_counter()
        if use_ref:
            otc = ref.Cloud(tp, None, tcov, tree=True, tree_threads=min(32, ncpu))
            osc = ref.Cloud(sp, None, scov, tree=False)
        else:
            otc = orc.Cloud(tp, None, tcov, tree=True)
            osc = orc.Cloud(sp, None, scov, tree=False)
        build_s = time.perf_counter() - t0
        best = None
        tried = {}
        for threads in counts:
            if use_ref:
                r = ref.align(otc, osc, ref.GICP, 1.0, 1.0, threads, args.cpu_iters, 0.0, 0.0)
            else:
                s = orc.default_setting(factor_kind=orc.GICP, num_threads=threads, max_iterations=args.cpu_iters, rotation_eps=0.0, translation_eps=0.0)
                r = orc.align(otc, osc, s)
            ips = (r.iterations + 1) / r.elapsed_sec
            tried[str(threads)] = ips
            if best is None or ips > best[0]:
                best = (ips, threads, r)
        ips, threads, r = best
        what = "unmodified reference code (oracle/_ref: registration_helper.cpp align, Registration<GICPFactor, ParallelReductionOMP>, built over an Eigen stand-in)" if use_ref else "oracle/ restatement of ParallelReductionOMP + KdTree + GICPFactor + LM"
        return {
            "value": ips,
            "unit": "iterations/s",
            "cores": threads,
            "kind": "reference" if use_ref else "port",
            "sample": "%s; full C3 pair (%d<->%d), %d outer LM iterations from identity, OpenMP schedule(guided,8); kd-tree build %.2fs excluded; iterations/s by thread count: %s"
            % (what, n, n, r.iterations + 1, build_s, json.dumps(tried)),
            "host_threads_available": ncpu,
        }
    except Exception as ex:  # noqa: BLE001
        return {"value": None, "unit": "iterations/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (ex,)}


if __name__ == "__main__":
    main()
