#!/usr/bin/env python3
"""bench.py — registration-iterations/s of the GICP hot path on MI355X (BASELINE.json metric, config C3).

  python bench.py --gpus N --steps K --warmup W            (N > 1 without WORLD_SIZE in the environment: spawns its own N ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (config.workload = "C3"): GICP with per-point covariances (k = 20), 1M target points <-> 1M source points, synthetic
planar scene (small_gicp_amd/synthetic.py, frozen in SURVEY.md §8d), max correspondence distance 1.0 m.
One STEP = one outer Levenberg-Marquardt iteration of Registration<GICPFactor>::align (registration/optimizer.hpp:100-144 of the
reference): 1x linearize (transform + exact NN + per-pair H/b/e + reduction) + the LM trial(s) (1x error pass each, normally one) +
the 6x6 solve on the host.  Registrations restart from the identity every 10 steps (and from a cold search state: no neighbour is
carried over from one registration to the next).  Inputs (clouds, covariances, search index) are resident in HBM before the timed
region.

Multi-GPU, --scaling strong (default): ONE 1M <-> 1M registration; the source cloud (registration/reduction_omp.hpp:32-58 is the
loop being partitioned) is split into N spatially contiguous shards, the target + its index are replicated, and the library
all-reduces ONE accumulator of 96 doubles per linearization (21 H, 6 b, e, inliers + the 63 moments of the quadratic error model)
with RCCL on its own stream (csrc/comm.hip); the trial errors of the LM steps are evaluated on every rank's host from the reduced
moments — no collective per error pass; every rank runs the same host LM on the reduced numbers.  value = iterations/s of that one
job.  Rank 0 also runs the unsharded registration and the bench asserts that the N-rank pose equals it (1e-9 in fp64 per-pair
arithmetic, 1e-5 in the timed fp32 arithmetic).  If the native communicator cannot be created the torch.distributed callback path (30 doubles per
linearization + 1 per error pass, still one rank per GPU, the all-reduce through torch's RCCL) is measured instead, the JSON says
"fallback": true with "fallback_reason", and stderr says so; --require-native makes that case fail instead.  --scaling weak: every rank owns an independent 1M-point source (value = N x the job's iteration rate).

Extra objects on the JSON line: "roofline" (K1 = search + factor kernel of one linearize pass, algorithmic bytes / HIP-event time vs
8 TB/s; traffic = FETCH_SIZE / WRITE_SIZE of one registration re-run under rocprofv3 when it is on the box), "cpu_baseline" (the
unmodified reference code, oracle/_ref, timed on this box's host cores on the same clouds; rank 0, N = 1 only), "parity_vs_reference"
(the same 10 LM iterations on the GPU against that reference run, with the number of differing correspondences), "to_convergence"
(whole registrations with the default termination criteria), "fp64" (the headline with fp64 per-pair math), "plane_icp_c2" (config
C2), "vgicp_c4", "kitti_odom" (C5, 100 scans; with N > 1 the scan's source points are sharded like the headline).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_POINT = {"linearize_gicp": 100, "error_gicp": 52, "linearize_plane_icp": 40,
                       # VGICP pass = the factor kernel alone: source point 16 + source covariance 32 + hash slot 8 + 4 + voxel mean 16 + voxel covariance 32 + correspondence 4
                       "linearize_vgicp": 112}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PROFILE_ALIGNS = 5  # registrations profiled pass by pass after the timed region (3 cold + 7 warm passes each on C3)
ITERS_PER_ALIGN = 10


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--points", type=int, default=1_000_000, help="points per cloud (C3 = 1M)")
    ap.add_argument("--neighbors", type=int, default=20)
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"], help="multi-GPU: strong = one job, source sharded; weak = one 1M source per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=10, help="outer LM iterations of the CPU baseline sample")
    ap.add_argument("--math", default="fp32", choices=["fp32", "fp64"])
    ap.add_argument("--odom-frames", type=int, default=100, help="frames of the KITTI-shaped scan-to-scan odometry leg (config C5, BASELINE.md: 100 scans); 0 = skip")
    ap.add_argument("--no-vgicp", action="store_true", help="skip the VGICP (config C4) leg")
    ap.add_argument("--no-plane", action="store_true", help="skip the point-to-plane (config C2) leg")
    ap.add_argument("--no-policy", action="store_true", help="skip the legs through the reference's Registration<> with ParallelReductionHIP + HipAligned<LM> (oracle/_ref/policy_bench)")
    ap.add_argument("--no-traffic", action="store_true", help="do not re-run one registration under rocprofv3 for the HBM traffic of K1")
    ap.add_argument("--require-native", action="store_true", help="N > 1: fail if the library's own RCCL communicator cannot be created (default: measure the torch.distributed callback path, flagged \"fallback\": true)")
    ap.add_argument("--allow-fallback", action="store_true", help="(the default since round 5; kept for old command lines)")
    ap.add_argument("--no-scaled", action="store_true", help="skip the headline on copies of the C3 clouds in other units of length (x0.01, x10)")
    ap.add_argument("--no-concurrent", action="store_true", help="skip the leg with 2 and 3 independent C3 registrations side by side on the one GPU")
    ap.add_argument("--no-preprocess", action="store_true", help="skip the per-stage roofline lines of the preprocessing kernels (voxel grid, index build, covariances)")
    ap.add_argument("--no-fp64", action="store_true", help="skip the fp64-math repetition of the headline")
    ap.add_argument("--sustain-s", type=float, default=3.0, help="extra (reported separately) sustained run of the same steps for this many seconds; 0 = skip")
    ap.add_argument("--force-dist", action="store_true", help="use the torch.distributed/RCCL path even at world size 1 (validation)")
    ap.add_argument("--oversubscribe", action="store_true", help="testing only: place ranks on device rank %% visible devices")
    return ap.parse_args()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under torch.distributed.run, one per GPU."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def morton_order(points):
    """Spatial (Morton) order of a cloud on the host: contiguous slices of it are compact source shards."""
    p = np.asarray(points, dtype=np.float64)
    lo, hi = p.min(0), p.max(0)
    q = np.clip(((p - lo) / np.maximum(hi - lo, 1e-9).max() * 1023.0), 0, 1023).astype(np.uint64)

    def spread(v):
        v = (v | (v << np.uint64(16))) & np.uint64(0x030000FF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x0300F00F)
        v = (v | (v << np.uint64(4))) & np.uint64(0x030C30C3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x09249249)
        return v

    key = spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2))
    return np.argsort(key, kind="stable")


def pose_error(A, B):
    E = np.linalg.inv(A) @ B
    return float(np.linalg.norm(E[:3, 3])), float(np.arccos(min(1.0, max(-1.0, (np.trace(E[:3, :3]) - 1) / 2))))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    # stdout carries the ONE JSON line and nothing else: RCCL / gloo print their banners to fd 1 from native code, so fd 1 is pointed
    # at stderr for the run and the JSON goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
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

        ndev = torch.cuda.device_count()
        if args.oversubscribe and ndev > 0:
            local_rank = local_rank % ndev
        if local_rank >= ndev:
            raise SystemExit("bench: rank %d needs GPU %d but only %d are visible" % (rank, local_rank, ndev))
        torch.cuda.set_device(local_rank)
        if args.oversubscribe:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    import small_gicp_amd as sga

    n = args.points
    strong = args.scaling == "strong"
    # ---- data: identical target on every rank; strong: the rank's shard of ONE source, weak: an independent source per rank ----
    T_gt = sga.synthetic.gt_transform()
    target = sga.synthetic.scene(n, 1)
    src_world = sga.synthetic.scene(n, 2 + (0 if strong else rank)).astype(np.float64)
    Ti = np.linalg.inv(T_gt)
    source = (src_world @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
    if strong and world > 1:
        source = source[morton_order(source)]

    if use_dist and not args.oversubscribe:
        stream = torch.cuda.current_stream().cuda_stream
        ctx = sga.Context(local_rank, stream=stream)
    else:
        ctx = sga.Context(local_rank if use_dist else 0)

    # ---- preprocessing on the GPU (untimed): covariances k = 20, search index, spatially sorted source ----
    t0 = time.perf_counter()
    tgt = sga.PointCloud(target, ctx=ctx)
    src_full = sga.PointCloud(source, ctx=ctx)
    sga.estimate_covariances(tgt, None, args.neighbors)
    sga.estimate_covariances(src_full, None, args.neighbors)  # over the whole source on every rank: a shard's covariances need its neighbours
    tree = sga.KdTree(tgt)
    if strong and world > 1:
        lo, hi = rank * n // world, (rank + 1) * n // world
        src = src_full.slice(lo, hi - lo)
    else:
        src = src_full
    problem = sga.Problem(tree, src, np.eye(4))
    ctx.synchronize()
    prep_s = time.perf_counter() - t0

    def setting_for(max_iters, math):
        return sga.make_setting("GICP", max_correspondence_distance=1.0, max_iterations=max_iters, rotation_eps=0.0, translation_eps=0.0, math_mode=math)

    # the unsharded registration of the same job (before the communicator exists): what the sharded run must reproduce
    # (fp64 per-pair arithmetic: the sharded sums then equal the unsharded ones to rounding of the fp64 row sums, 1e-9 m; in the
    # default fp32 arithmetic the partial sums GROUP by shard — 64 / 256 points per fp32 sum — and the poses agree to ~1e-6 m)
    ref_pose = {}
    if strong and world > 1 and rank == 0:
        for m in ("fp64", "fp32"):
            ref_pose[m] = sga.Problem(tree, src_full, np.eye(4)).align(setting_for(ITERS_PER_ALIGN, m)).T_target_source

    native_comm = False
    transport = None
    fallback_reason = None
    if use_dist:
        # native path: the library all-reduces its accumulators with RCCL on its own stream (no Python in the iteration loop);
        # torch.distributed only carries the 128-byte communicator id, the barriers and the max-over-ranks of the wall time
        try:
            if args.oversubscribe:
                # testing on fewer devices than ranks: the SAME protocol (one 96-double all-reduce per linearization inside the library,
                # per-rank error model) with the sum carried by gloo through sga_comm_init_callback instead of RCCL
                def _host_allreduce(values):
                    t = torch.from_numpy(values)
                    dist.all_reduce(t)
                    return t.numpy()

                ctx.comm_init_callback(world, rank, _host_allreduce)
                transport = "host callback over gloo (sga_comm_init_callback; --oversubscribe testing mode)"
            else:
                if os.environ.get("SGA_BENCH_FORCE_FALLBACK"):  # exercises the branch below on a box where the communicator would work
                    raise RuntimeError("SGA_BENCH_FORCE_FALLBACK is set")
                ids = [sga.Context.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(ids, src=0)
                ctx.comm_init(world, rank, ids[0])
                transport = "native ncclAllReduce (RCCL) on the library stream"
            native_comm = True
        except Exception as ex:  # noqa: BLE001
            if args.require_native and not args.oversubscribe:
                raise SystemExit("bench: native RCCL communicator unavailable (%r) and --require-native was given" % (ex,))
            fallback_reason = repr(ex)
            print("bench: native RCCL communicator unavailable (%r); MEASURING THE torch.distributed CALLBACK PROTOCOL instead (\"fallback\": true in the JSON line)" % (ex,), file=sys.stderr)

    host_tensors = args.oversubscribe  # gloo fallback reduces on the host
    if use_dist and not native_comm:
        acc = torch.zeros(sga._lib.ACCUM_DOUBLES, dtype=torch.float64, device="cuda")
        acc1 = torch.zeros(1, dtype=torch.float64, device="cuda")

        def reduce_(t):
            if not host_tensors:
                dist.all_reduce(t)
                return t.cpu().numpy()
            h = t.cpu()
            dist.all_reduce(h)
            return h.numpy()

        cur_math = [args.math]

        def lin_cb(T):
            problem.linearize_async(setting_for(1, cur_math[0]).factor, T, acc.data_ptr())
            ctx.synchronize()
            return sga.unpack_accumulator(reduce_(acc))

        def err_cb(T):
            problem.error_async(setting_for(1, cur_math[0]).factor, T, acc1.data_ptr())
            ctx.synchronize()
            return float(reduce_(acc1)[0])

        def run_align(max_iters, math=args.math):
            cur_math[0] = math
            return sga.optimize(setting_for(max_iters, math), np.eye(4), lin_cb, err_cb)

    else:

        def run_align(max_iters, math=args.math):
            return problem.align(setting_for(max_iters, math), np.eye(4))

    def run_steps(k, math=args.math):
        done = 0
        last = None
        while done < k:
            last = run_align(min(ITERS_PER_ALIGN, k - done), math)
            done += last.iterations + 1
        return done, last

    def barrier():
        if use_dist:
            dist.barrier()
            if not args.oversubscribe:
                torch.cuda.synchronize()
        ctx.synchronize()

    def timed(k, math=args.math):
        barrier()
        t0 = time.perf_counter()
        done, last = run_steps(k, math)
        barrier()
        el = time.perf_counter() - t0
        if use_dist:
            tmax = torch.tensor([el], dtype=torch.float64, device="cpu" if host_tensors else "cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.cpu()[0])
        return done, last, el

    run_steps(args.warmup)
    # The timed region (it carries no event records at all).  A request for fewer than 100 steps is a region of a few milliseconds (the
    # driver's --steps 20: two registrations, 2.5 ms) whose rate scatters by several per cent from run to run; such a region is therefore
    # timed REPEATS = 9 times back to back — each one exactly `steps` steps between the barriers — and `value` / `ms_per_step` are those of
    # the MEDIAN region; every region is listed ("median of >= 5 runs after 1 warm-up").
    REPEATS = 9 if args.steps < 100 else 1
    regions = [timed(args.steps) for _ in range(REPEATS)]
    order = sorted(range(REPEATS), key=lambda i: regions[i][2] / regions[i][0])
    steps_done, last, elapsed = regions[order[REPEATS // 2]]
    timed_regions = [{"steps": d, "seconds": e, "iterations_per_s": d / e * (1 if strong else world)} for d, _, e in regions]
    # the kernel times of the roofline come from PROFILE_ALIGNS further, untimed registrations with HIP events around EVERY pass
    # (>= 10 cold and >= 20 warm launches instead of the three a sampled timed region gave)
    ctx.set_profiling(1)
    for _ in range(PROFILE_ALIGNS):
        run_align(ITERS_PER_ALIGN)
    kms = ctx.kernel_ms()
    stats = problem.pass_stats()
    ctx.set_profiling(False)

    # sanity: the pose the timed registrations converge to
    pose_err_t, pose_err_r = pose_error(last.T_target_source, T_gt)

    shard_check = None
    if strong and world > 1:
        shard_check = {}
        for m, tol in (("fp64", 1e-9), ("fp32", 1e-5)):
            full = run_align(ITERS_PER_ALIGN, m)  # a complete registration of the sharded job
            if rank == 0:
                dt, dr = pose_error(full.T_target_source, ref_pose[m])
                shard_check[m] = {"dt_m": dt, "dr_rad": dr, "tolerance": tol}
                assert dt <= tol and dr <= tol, "sharded registration differs from the unsharded one: %r" % (shard_check,)
        if rank == 0:
            shard_check["note"] = "fp64 per-pair arithmetic: equal to the rounding of the fp64 row sums; fp32 (the timed mode): the fp32 partial sums group by shard"

    sustained = None
    if args.sustain_s > 0:
        k = max(args.steps, int(args.sustain_s * steps_done / max(elapsed, 1e-6)))
        d2, _, e2 = timed(k)
        sustained = {"steps": d2, "seconds": e2, "iterations_per_s": d2 / e2 * (1 if strong else world)}

    fp64 = None
    if not args.no_fp64 and args.math == "fp32":
        run_steps(ITERS_PER_ALIGN, "fp64")
        d3, l3, e3 = timed(max(40, min(args.steps, 200)), "fp64")
        t3, r3 = pose_error(l3.T_target_source, T_gt)
        fp64 = {"iterations_per_s": d3 / e3 * (1 if strong else world), "ms_per_step": 1e3 * e3 / d3, "steps": d3, "final_pose_error": {"trans_m": t3, "rot_rad": r3},
                "note": "the same steps with fp64 per-pair arithmetic (SGA_MATH_FP64); data in HBM stays fp32"}

    per_rank = None
    if use_dist:  # so that the first real multi-GPU run explains itself: every rank's pass time and its time inside the collective
        mine = {"rank": rank, "source_points": src.size(), "k1_avg_us": kms["linearize_ms"] * 1e3, "cold_pass_avg_us": kms["cold_ms"] * 1e3, "warm_pass_avg_us": kms["warm_ms"] * 1e3,
                "collective_avg_us": kms["comm_ms"] * 1e3, "collective_launches_timed": kms["comm_calls"], "passes_timed": kms["linearize_calls"]}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    out = None
    if rank == 0:
        job_rate = steps_done / elapsed
        value = job_rate * (1 if strong else world)
        lin_us = kms["linearize_ms"] * 1e3
        err_us = kms["error_ms"] * 1e3
        # the search + factor kernel of the cold passes alone: (all passes' - the warm passes') kernel time / cold launches
        cold_search_us = (kms["search_ms"] * kms["search_calls"] - kms["warm_search_ms"] * kms["warm_calls"]) / max(kms["cold_calls"], 1) * 1e3
        n_rank = src.size()
        achieved = (ALG_BYTES_PER_POINT["linearize_gicp"] * n_rank) / (lin_us * 1e-6) / 1e9 if lin_us > 0 else None
        traffic, traffic_source, traffic_measured = None, None, False
        if world == 1 and not use_dist:
            traffic, traffic_source = measure_traffic(args)
            traffic_measured = traffic is not None
        if traffic is None:
            why = traffic_source
            tpath = os.path.join(ROOT, "profiles", "k1_traffic.json")
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_source = "NOT measured in this run (%s): profiles/k1_traffic.json, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/profile_gpu.sh at commit %s" % (why, tj.get("commit", "?"))
                except Exception:  # noqa: BLE001
                    traffic = None
        out = {
            "metric": "GICP iterations/sec (1M<->1M pts)",
            "value": value,
            "unit": "iterations/s",
            "n_gpus": world,
            "fallback": bool(use_dist and not native_comm),
            "fallback_reason": fallback_reason,
            "steps": steps_done,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / steps_done,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32" if args.math == "fp32" else "f64",
            "data": "synthetic",
            "config": {
                "workload": "C3: GICP, per-point covariances k=20, %d target <-> %d source points%s, max_corr_dist 1.0 m" % (n, n, "" if strong else " per GPU"),
                "step": "1 outer LM iteration = one linearize pass (search + factor kernel, which also accumulates the quadratic error model) + host 6x6 solve(s) + the trial errors evaluated on the host from that model (exact for the cached correspondences; replaces the reference's error passes); restart from identity (and a cold search state) every %d steps" % ITERS_PER_ALIGN,
                "parallelism": ("%s scaling: source %s x%d, target index replicated, %s"
                                % (args.scaling, "sharded (contiguous Morton ranges of one cloud)" if strong else "one independent cloud per rank", world,
                                   "all-reduce of 96 doubles per linearize (the system + the error-model moments): %s" % transport if native_comm
                                   else "FALLBACK (no RCCL communicator): torch.distributed all-reduce of 30 doubles per linearize + 1 per error pass through sga_linearize_async / sga_error_async callbacks")) if use_dist else "single GPU",
                "source_points_total": n * (1 if strong else world),
                "source_points_per_gpu": n_rank,
            },
            "iters_per_sec_job": job_rate,
            "timed_regions": {"repeats": REPEATS, "reported": "median" if REPEATS > 1 else "the one region", "regions": timed_regions,
                              "note": "fewer than 100 steps requested: the region of exactly `steps` steps is timed %d times back to back and value / ms_per_step are the median region's" % REPEATS if REPEATS > 1 else "one timed region of `steps` steps"},
            "roofline": {
                "kernel": "K1 = the GPU side of one linearize pass: search_linearize_kernel<float, GICP> (cold passes and warm passes after larger motions: every search wave also evaluates the factors of its 64 points, moment form) or nn_search_queue_kernel<float, warm, GICP> (warm passes after small motions: certificate check, queue-fed walks, factors per chunk), followed by reduce_rows_kernel (fp64 sum of the partial rows); average over the passes of whole registrations (HIP events around the launches)",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                "traffic": traffic,
                "traffic_measured": traffic_measured,
                "traffic_source": traffic_source,
                "alg_bytes_per_launch": ALG_BYTES_PER_POINT["linearize_gicp"] * n_rank,
                "avg_launch_us": lin_us,
                "launches_timed": kms["linearize_calls"],
                "launches_timed_note": "HIP events around every pass of %d registrations run right after the timed region (the timed region itself carries no events)" % PROFILE_ALIGNS,
                "frac_dominant_kernel": ((ALG_BYTES_PER_POINT["linearize_gicp"] * n_rank) / (cold_search_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if cold_search_us > 0 else None,
                "dominant_kernel": "search_linearize_kernel<float, GICP, cold>: the search + factor kernel of a cold pass alone (no reduce), %.1f us on average over %d launches" % (cold_search_us, kms["cold_calls"]),
                "search_and_factor_kernel_avg_us": kms["search_ms"] * 1e3,
                "reduce_rows_and_launch_gap_avg_us": (kms["linearize_ms"] - kms["search_ms"]) * 1e3,
                "cold_pass_avg_us": kms["cold_ms"] * 1e3,
                "cold_passes_timed": kms["cold_calls"],
                "warm_pass_avg_us": kms["warm_ms"] * 1e3,
                "warm_pass_search_avg_us": kms["warm_search_ms"] * 1e3,
                "warm_passes_timed": kms["warm_calls"],
                "pass_stats": stats,
                "error_kernel_avg_us": err_us if err_us > 0 else None,  # None: no error pass ran (the error model answered)
                "error_kernel_achieved_GBs": (ALG_BYTES_PER_POINT["error_gicp"] * n_rank) / (err_us * 1e-6) / 1e9 if err_us > 0 else None,
            },
            "preprocess_s": prep_s,
            "final_pose_error": {"trans_m": pose_err_t, "rot_rad": pose_err_r},
        }
        if MEASURED_VALU:
            # What actually binds K1: the walks are VALU work at a third to a half of the lanes.  A wave64
            # instruction occupies a 16-lane SIMD for 4 cycles; 256 CUs x 4 SIMDs at 2.4 GHz.
            insts = MEASURED_VALU["insts_per_launch"]
            floor_us = insts * 4.0 / (256 * 4) / 2400.0
            out["roofline"]["valu_issue"] = {
                "wave_instructions_per_launch": insts, "floor_us": floor_us, "frac_of_issue_slots": floor_us / lin_us if lin_us > 0 else None,
                "floor_us_at_2_0_ghz": insts * 4.0 / (256 * 4) / 2000.0,
                "note": "(floor_us is at the 2.4 GHz maximum clock; MI355X_MICROARCH.md, DVFS: dense vector code sustains 1.9 - 2.3 GHz, which raises the floor accordingly.)  SQ_INSTS_VALU of the search + factor + reduce kernels per pass (rocprofv3 --pmc, its own pass over one more registration of %d passes) x 4 cycles / 1024 SIMDs / 2.4 GHz: "
                        "the time the pass would take if every SIMD issued a vector instruction every cycle it can; frac = that floor / avg_launch_us" % MEASURED_VALU["passes"]}
        if per_rank is not None:
            out["per_rank"] = per_rank
            out["per_rank_note"] = ("k1 = search + factors + row reduction of the rank's shard (HIP events); collective = from the end of the row reduction to the end of the all-reduce of the 96-double "
                                    "accumulator on the same stream, i.e. the collective's own latency PLUS the wait for the slowest rank")
        if shard_check is not None:
            out["sharded_vs_unsharded"] = shard_check
        if sustained is not None:
            out["sustained"] = sustained
        if fp64 is not None:
            out["fp64"] = fp64
        single = world == 1 and not use_dist
        if single:
            out["to_convergence"] = convergence_leg(problem, setting_for, problem.pass_stats, ctx, T_gt)
        if single and not args.no_cpu_baseline:
            out["cpu_baseline"], ref_result, ref_nn = cpu_baseline(sga, tgt, src, n, args)
            if ref_result is not None:
                out["parity_vs_reference"] = parity_vs_reference(problem, ref_result, setting_for(args.cpu_iters, "fp32"), setting_for(args.cpu_iters, "fp64"), out["cpu_baseline"]["kind"], ref_nn)
            ref_nn = None  # releases the CPU clouds
        if single and not args.no_policy:
            out["policy_c3"] = policy_leg(sga, "GICP", tgt, src, value, last.T_target_source)
        if single and not args.no_plane:
            out["plane_icp_c2"] = plane_icp_leg(sga, ctx, args)
            if not args.no_policy and isinstance(out["plane_icp_c2"], dict) and out["plane_icp_c2"].get("value"):
                out["policy_c2"] = policy_leg(sga, "PLANE_ICP", None, None, out["plane_icp_c2"]["value"], None, n=100_000)
        if single and not args.no_vgicp:
            out["vgicp_c4"] = vgicp_leg(sga, ctx, tgt, src, args)
            if not args.no_policy and "value" in out["vgicp_c4"]:
                # the same VGICP through the reference's Registration<GICPFactor, ParallelReductionHIP, ..., HipAligned<LM>>::align(voxelmap, source, voxelmap):
                # the target is the reference's own GaussianVoxelMap object, built on the host by the reference's insert()
                out["policy_c4"] = policy_leg(sga, "VGICP", tgt, src, out["vgicp_c4"]["value"], None)
        if single and not args.no_concurrent:
            out["concurrent_registrations"] = concurrent_leg(sga, tree, src, args)
        if single and not args.no_scaled:
            out["scaled_scenes"] = scaled_scenes_leg(sga, ctx, target, source, args, value)
        if single and not args.no_preprocess:
            out["preprocess_rooflines"] = preprocess_rooflines(sga, ctx, target, args)
            out["fresh_align_c3"] = fresh_align_leg(sga, ctx, target, source, args)
        if single and args.odom_frames > 1:
            out["kitti_odom"] = odometry_leg(sga, args, None)
        if single and not args.no_policy:
            out["helper_c1"] = helper_leg(sga, "c1")
            if args.odom_frames > 1:
                out["helper_c5"] = helper_leg(sga, "c5")
        if world > 1:
            out["scaling_model"] = scaling_model(world, per_rank)
    if use_dist and native_comm and world > 1 and args.odom_frames > 1:
        r = odometry_leg(sga, args, (rank, world, ctx))
        # the other way to spread C5 over the GPUs: whole frame pairs per rank, no collective
        pairs = None
        try:
            from small_gicp_amd import odometry

            pr = odometry.run_synthetic_pairs(args.odom_frames, rank, world, device=local_rank)
            t = torch.tensor([pr["seconds"]], dtype=torch.float64, device="cpu" if host_tensors else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            pairs = {"ms_per_scan": 1e3 * float(t.cpu()[0]) / args.odom_frames, "unit": "ms/scan (total: voxel grid + index + covariances + align)", "frames_per_rank": pr["frames"],
                     "mode": "frame pairs are independent under the reference's protocol (identity initial guess per pair): contiguous blocks of frames per rank, every rank preprocesses its block + 1 scan, no collective"}
        except Exception as ex:  # noqa: BLE001
            pairs = {"error": repr(ex)}
        # ... and each rank's block as a flow of stages (2 preprocessing x 2 registration workers per GPU)
        pairs_flow = None
        try:
            pf = odometry.run_synthetic_pairs(args.odom_frames, rank, world, device=local_rank, flow=(2, 2))
            t = torch.tensor([pf["seconds"]], dtype=torch.float64, device="cpu" if host_tensors else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            worst = 0.0
            if "pr" in dir() and isinstance(pr, dict):
                worst = max([float(np.abs(pf["relative_poses"][f] - pr["relative_poses"][f]).max()) for f in pf["relative_poses"] if f in pr["relative_poses"]] or [0.0])
            pairs_flow = {"ms_per_scan": 1e3 * float(t.cpu()[0]) / args.odom_frames, "unit": "ms/scan (total)", "frames_per_rank": pf["frames"], "workers_per_rank": "2 preprocessing x 2 registration",
                          "max_abs_relative_pose_difference_vs_frame_by_frame": worst}
        except Exception as ex:  # noqa: BLE001
            pairs_flow = {"error": repr(ex)}
        if rank == 0:
            out["kitti_odom"] = r
            out["kitti_odom_frame_pairs_per_rank"] = pairs
            out["kitti_odom_frame_pairs_per_rank_flow"] = pairs_flow
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def helper_leg(sga, which):
    """ms per call of the reference's HELPER API (registration/registration_helper.hpp) served by integration/registration_helper_hip.cpp
    (oracle/_ref/test_helper_hip: the reference-side binding compiled against the unmodified reference headers — not the oracle):
    align(points ...) = raw points up, result down, nothing else crosses PCIe (downsampling 0.25 m, k = 10, GICP); align(clouds + tree) on
    clouds that came out of preprocess_points = their device twins, no upload; preprocess_points = device pipeline + download + the host
    KdTree its return type promises.  c1 = data/target.ply <-> source.ply (69k points each); c5 = two consecutive KITTI-shaped scans (~125k points each)."""
    import subprocess
    import tempfile

    binary = os.path.join(ROOT, "oracle", "_ref", "test_helper_hip")
    if not os.path.exists(binary):
        return {"skipped": "oracle/_ref/test_helper_hip is not built (make -C oracle/ref where /root/reference is mounted)"}
    try:
        if which == "c1":
            d = np.load(os.path.join(ROOT, "tests", "golden", "c1_points.npz"))
            clouds = (d["target"][:, :3], d["source"][:, :3])
        else:
            clouds = (sga.synthetic.kitti_like_scan(0)[0][:, :3], sga.synthetic.kitti_like_scan(1)[0][:, :3])
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            paths = []
            for name, a in zip(("target", "source"), clouds):
                paths.append(os.path.join(tmp, name + ".bin"))
                np.ascontiguousarray(a, dtype="<f4").tofile(paths[-1])
            p = subprocess.run([binary, paths[0], paths[1], "20"], capture_output=True, text=True, timeout=600)
        for ln in p.stdout.splitlines():
            if ln.startswith("TIMING "):
                r = json.loads(ln[7:])
                r["unit"] = "ms per call, 20 calls after the first"
                return r
        return {"error": "no TIMING line: rc %d %s" % (p.returncode, p.stderr[-300:])}
    except Exception as ex:  # noqa: BLE001
        return {"error": repr(ex)}


def scaling_model(world, per_rank):
    """What the design predicts for ONE 1M <-> 1M registration sharded over N GPUs (profiles/r06_shard_model.txt: one GPU
    doing the work of one rank of N): K1 per pass of a 1 / N source slice + 6.5 us of host work + the all-reduce.  Printed beside the
    measurement so that the first real multi-GPU run is a one-line comparison; `with_measured_collective` replaces the assumed 20 us by
    this run's own per_rank.collective_avg_us."""
    k1 = {1: 118.0, 2: 83.2, 4: 60.7, 8: 52.2}  # us per pass, measured on the round's final build (scripts/diag_shards.py)
    host_us, assumed = 6.5, 20.0
    n = min(k1, key=lambda g: abs(g - world))
    out = {"shard_k1_us": k1, "host_us": host_us, "assumed_allreduce_us": assumed,
           "predicted_iterations_per_s": {str(g): 1e6 / (v + host_us + (assumed if g > 1 else 0.0)) for g, v in k1.items()},
           "note": "a pass over an N-th of the source still costs half a pass (search chains, launch, row reduction, hand-off do not shrink with the shard): one job saturates near 1.6x at 8 GPUs; "
                   "weak scaling (--scaling weak) and frame pairs per rank (kitti_odom_frame_pairs_per_rank) are the modes that scale"}
    try:
        coll = [r.get("collective_avg_us") for r in (per_rank or []) if r and r.get("collective_avg_us")]
        if coll:
            out["with_measured_collective"] = {"collective_avg_us": max(coll), "predicted_iterations_per_s": 1e6 / (k1[n] + host_us + max(coll))}
    except Exception:  # noqa: BLE001
        pass
    return out


def policy_leg(sga, kind, tgt, src, cabi_rate, cabi_pose, n=None):
    """The headline THROUGH the boundary north_star names: the same clouds handed to the reference's own
    Registration<Factor, ParallelReductionHIP, NullFactor, DistanceRejector, HipAligned<LevenbergMarquardtOptimizer>>::align (the
    unmodified reference headers, oracle/_ref/policy_bench), 10 fixed LM iterations per align like the headline.  `policy_calls` =
    iterations / time inside the policy's linearize() + error() during the align bracket: the rate a reference user gets for the
    hot path; the rest of an align() is the reference's own host code (its factor vector, its count over the host factors)."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import policy_bench

        clouds = None
        if tgt is not None:
            clouds = (tgt.xyz(), src.xyz(), sga.api.sym6_from_mats(tgt.covs()), sga.api.sym6_from_mats(src.covs()))
        r = policy_bench.run(kind, n or len(clouds[0]), reps=15 if kind in ("GICP", "VGICP") else 20, clouds=clouds)
        if r is None:
            return {"error": "oracle/_ref/policy_bench did not travel with the repository (built where /root/reference is mounted: make -C oracle/ref)"}
        out = {k: r[k] for k in ("points", "whole_align_iterations_per_s", "whole_align_median_iterations_per_s", "align_ms", "inside_the_optimizer_iterations_per_s", "policy_calls_iterations_per_s", "per_align_ms", "lean",
                                 "reduction_slot_only_iterations_per_s", "first_align_s", "first_bind_s", "num_inliers")}
        out["c_abi_iterations_per_s"] = cabi_rate
        out["policy_calls_over_c_abi"] = r["policy_calls_iterations_per_s"] / cabi_rate if cabi_rate else None
        if cabi_pose is not None:
            T = np.array(r["T"]).reshape(4, 4).T
            dt, dr = pose_error(T, cabi_pose)
            out["pose_vs_c_abi"] = {"trans_m": dt, "rot_rad": dr}
        out["note"] = ("whole_align = iterations / wall time of the timed align() calls (their mean; align_ms lists every one of them and whole_align_median is the rate of the median call: on a two-socket host the "
                       "first calls of a run sometimes take 2 - 5 ms — the content check streams 320 MB of host memory beside the registration and waits for the operating system's page migration) incl. the reference's own std::vector<Factor>(n) (registration.hpp:41: 144 B per source point, per_align_ms.reference_factor_vector) and its "
                       "count over the host factors (optimizer.hpp:146); inside_the_optimizer = the reference's optimize() between begin_align and end_align; policy_calls = inside ParallelReductionHIP::linearize / error only; "
                       "lean = verify_content and sync_inliers off; reduction_slot_only = Registration<Factor, ParallelReductionHIP> without HipAligned (since the Registration<> specialisation: the same bracket; before: content check + factor fill per linearize); upload + index build: first_bind_s, once per cloud")
        return out
    except Exception as ex:  # noqa: BLE001
        return {"error": repr(ex)}


def parity_vs_reference(problem, ref, st32, st64, kind, ref_nn=None):
    """The GPU against the CPU run of the same registration (same clouds and covariances, identity start, the same fixed number of LM
    iterations): pose, iteration count, inliers, final H / error — and, at the reference's final pose, the number of source points whose
    correspondence differs from the reference's own nearest_neighbor_search + DistanceRejector (ref_nn(T) -> target index or -1 per
    source point in the caller's order)."""
    out = {}
    for name, st in (("fp32", st32), ("fp64", st64)):
        r = problem.align(st, np.eye(4))
        dt, dr = pose_error(r.T_target_source, ref.T_target_source)
        out[name] = {
            "dt_m": dt,
            "dr_rad": dr,
            "iterations": [int(r.iterations), int(ref.iterations)],
            "num_inliers": [int(r.num_inliers), int(ref.num_inliers)],
            "inlier_delta": int(r.num_inliers) - int(ref.num_inliers),
            "rel_err_H": float(np.abs(r.H - ref.H).max() / np.abs(ref.H).max()),
            "rel_err_e": float(abs(r.error - ref.error) / abs(ref.error)),
        }
        if ref_nn is not None:
            try:
                want = ref_nn(ref.T_target_source)
                problem.linearize(st.factor, ref.T_target_source)
                got = problem.factors()[0]
                out[name]["correspondences_differing"] = int((got != want).sum())
                out[name]["correspondences_total"] = int(len(want))
            except Exception as ex:  # noqa: BLE001
                out[name]["correspondences_error"] = repr(ex)
    out["against"] = ("oracle/_ref: registration_helper.cpp align() = Registration<GICPFactor, ParallelReductionOMP> of the unmodified reference (double), registration_helper.cpp:81-137; "
                      "correspondences: KdTree::nearest_neighbor_search + DistanceRejector of the same build at the reference's final pose"
                      if kind == "reference" else "oracle/ restatement (the compiled reference did not travel with the repository)")
    return out


def convergence_leg(problem, setting_for, stats_fn, ctx, T_gt, reps=5):
    """Whole registrations with the DEFAULT termination criteria (rotation_eps 0.1 deg, translation_eps 1e-3 m,
    registration/termination_criteria.hpp) from the identity and a cold search state: what a caller of align() pays.  (The headline
    protocol runs 10 LM iterations with eps = 0, so half of its passes are post-convergence passes over settled certificates.)"""
    import small_gicp_amd as sga

    st = sga.make_setting("GICP", max_correspondence_distance=1.0, max_iterations=20, math_mode="fp32")
    problem.align(st, np.eye(4))
    s0 = stats_fn()
    ctx.synchronize()
    t0 = time.perf_counter()
    iters = 0
    last = None
    for _ in range(reps):
        last = problem.align(st, np.eye(4))
        iters += last.iterations + 1
    ctx.synchronize()
    el = time.perf_counter() - t0
    s1 = stats_fn()
    dt, dr = pose_error(last.T_target_source, T_gt)
    return {"iterations_per_s": iters / el, "ms_per_registration": 1e3 * el / reps, "mean_iterations": iters / reps, "converged": bool(last.converged),
            "passes_per_registration": {"cold": (s1["cold_passes"] - s0["cold_passes"]) / reps, "warm": (s1["warm_passes"] - s0["warm_passes"]) / reps},
            "final_pose_error": {"trans_m": dt, "rot_rad": dr},
            "note": "default termination criteria (0.1 deg / 1e-3 m), max_iterations 20, identity start, cold search state per registration"}


def plane_icp_leg(sga, ctx, args):
    """Config C2 (BASELINE.json configs[1]): point-to-plane ICP (factors/plane_icp_factor.hpp:19-57), 100k target <-> 100k source points
    of the same synthetic scene, target normals k = 20; same step definition as the headline.  Algorithmic bytes: 40 B per source point
    (p_s 12 + matched p_t 12 + n_t 12 read, index 4 written)."""
    try:
        n = 100_000
        T_gt = sga.synthetic.gt_transform()
        target, source, _ = sga.synthetic.registration_pair(n)
        tgt, src = sga.PointCloud(target, ctx=ctx), sga.PointCloud(source, ctx=ctx)
        sga.estimate_normals(tgt, None, 20)
        tree = sga.KdTree(tgt)
        problem = sga.Problem(tree, src, np.eye(4))
        s = sga.make_setting("PLANE_ICP", max_correspondence_distance=1.0, max_iterations=ITERS_PER_ALIGN, rotation_eps=0.0, translation_eps=0.0, math_mode=args.math)
        for _ in range(2):
            problem.align(s, np.eye(4))
        ctx.set_profiling(7)  # HIP events around every 7th pass (coprime with the 10 passes of a registration)
        ctx.synchronize()
        t0 = time.perf_counter()
        steps = 0
        last = None
        while steps < 200:
            last = problem.align(s, np.eye(4))
            steps += last.iterations + 1
        ctx.synchronize()
        el = time.perf_counter() - t0
        kms = ctx.kernel_ms()
        ctx.set_profiling(False)
        lin_us = kms["linearize_ms"] * 1e3
        dt, dr = pose_error(last.T_target_source, T_gt)
        return {"value": steps / el, "unit": "iterations/s", "ms_per_step": 1e3 * el / steps, "k1_avg_us": lin_us, "cold_pass_avg_us": kms["cold_ms"] * 1e3, "warm_pass_avg_us": kms["warm_ms"] * 1e3,
                "alg_bytes_per_point": ALG_BYTES_PER_POINT["linearize_plane_icp"], "achieved_GBs": (ALG_BYTES_PER_POINT["linearize_plane_icp"] * n) / (lin_us * 1e-6) / 1e9 if lin_us > 0 else None,
                "frac_of_hbm_peak": (ALG_BYTES_PER_POINT["linearize_plane_icp"] * n) / (lin_us * 1e-6) / 1e9 / HBM_PEAK_GBS if lin_us > 0 else None,
                "final_pose_error": {"trans_m": dt, "rot_rad": dr},
                "workload": "C2: point-to-plane ICP, %d target <-> %d source points, target normals k=20, max_corr_dist 1.0 m; 10 LM iterations from identity per registration" % (n, n)}
    except Exception as ex:  # noqa: BLE001
        return {"error": repr(ex)}


MEASURED_VALU = {}  # filled by measure_traffic: wave-level VALU instructions of K1 per pass (SQ_INSTS_VALU)


def measure_traffic(args):
    """HBM bytes of K1 per pass, measured: ONE more C3 registration (scripts/one_registration.py: the same clouds, 10 LM iterations) under
    rocprofv3 --kernel-trace --pmc FETCH_SIZE, and again with --pmc WRITE_SIZE (separate passes, kernel-trace only, as
    MI355X_MICROARCH.md prescribes); FETCH_SIZE x2 on gfx950 (128-B requests tallied at 64 B), WRITE_SIZE as reported; both in KB.
    Returns (bytes per pass or None, description)."""
    import csv
    import glob
    import shutil
    import tempfile

    exe = shutil.which("rocprofv3")
    if exe is None or args.no_traffic:
        return None, "rocprofv3 not on this box" if exe is None else "--no-traffic"
    tmp = tempfile.mkdtemp(prefix="sga_traffic_")
    totals = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
            d = os.path.join(tmp, ctr)
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable, os.path.join(ROOT, "scripts", "one_registration.py"), str(args.points)]
            p = subprocess.run(cmd, cwd=tmp, env=dict(os.environ, TMPDIR=tmp), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
            if p.returncode != 0:
                if ctr == "SQ_INSTS_VALU":
                    break  # the instruction count is extra information
                return None, "rocprofv3 %s pass failed: %s" % (ctr, p.stderr.decode(errors="replace")[-300:])
            kb, passes = 0.0, 0
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r.get("Counter_Name") != ctr:
                        continue
                    kn = r.get("Kernel_Name", "")
                    # the kernels of a pass: search_linearize / nn_search_queue / nn_search / certify_linearize / linearize / grid_* and
                    # reduce_rows_kernel, which runs exactly once per pass and therefore counts the passes
                    if "linearize_kernel" in kn or "nn_search" in kn or "grid_ring1" in kn or "grid_finish" in kn or "reduce_rows_kernel" in kn:
                        kb += float(r.get("Counter_Value", 0))
                        if "reduce_rows_kernel" in kn:
                            passes += 1
            if passes == 0:
                if ctr == "SQ_INSTS_VALU":
                    break
                return None, "no K1 dispatch in the %s pass" % ctr
            totals[ctr] = (kb / passes, passes)
        hbm = int((2.0 * totals["FETCH_SIZE"][0] + totals["WRITE_SIZE"][0]) * 1024)
        MEASURED_VALU.clear()
        if "SQ_INSTS_VALU" in totals:
            MEASURED_VALU.update(insts_per_launch=totals["SQ_INSTS_VALU"][0], passes=totals["SQ_INSTS_VALU"][1])
        return hbm, ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) around one more C3 registration (%d passes); FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B), "
                     "WRITE_SIZE as reported, KB; search + factor + reduce kernels of a pass" % totals["FETCH_SIZE"][1])
    except Exception as ex:  # noqa: BLE001
        return None, "traffic measurement failed: %r" % (ex,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


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
        ctx.set_profiling(7)  # HIP events around every 7th pass (coprime with the 10 passes of a registration)
        ctx.synchronize()
        t0 = time.perf_counter()
        steps = 0
        while steps < 100:
            steps += problem.align(s, np.eye(4)).iterations + 1
        ctx.synchronize()
        el = time.perf_counter() - t0
        kms = ctx.kernel_ms()
        ctx.set_profiling(False)
        lin_us = kms["linearize_ms"] * 1e3
        gbs = ALG_BYTES_PER_POINT["linearize_vgicp"] * src.size() / (lin_us * 1e-6) / 1e9 if lin_us > 0 else None  # whole pass (kernel + row reduction + launch gap); the kernel alone: profiles/*_c4_c2_kernel_stats.csv
        return {"value": steps / el, "unit": "iterations/s", "ms_per_step": 1e3 * el / steps, "voxelmap_build_s": build_s, "linearize_kernel_avg_us": lin_us, "error_kernel_avg_us": kms["error_ms"] * 1e3,
                "alg_bytes_per_point": ALG_BYTES_PER_POINT["linearize_vgicp"], "achieved_GBs": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS if gbs else None,
                "workload": "C4: VGICP, GaussianVoxelMap(0.5 m) of the 1M-point target, 1M source points"}
    except Exception as ex:  # noqa: BLE001
        return {"error": repr(ex)}


def concurrent_leg(sga, tree, src, args):
    """J independent C3 registrations side by side on the ONE GPU: one context (= one stream) and one host thread per job, the target index
    and the source cloud shared, a factor state (sga_problem) per job.  A lone registration leaves the machine partly idle — the drain of
    its cold passes, the launch gaps, the row reduction, the host's 6x6 solve — and other jobs fill that.  NOT the headline
    (`value` is the rate of one registration, as in every round): the throughput a server with several scan pairs in flight would see."""
    import sys
    import threading

    out = {"jobs": {}}
    old_switch = sys.getswitchinterval()
    try:
        sys.setswitchinterval(2e-5)  # the threads live inside ctypes calls (GIL released) and need it for microseconds in between
        st = sga.make_setting("GICP", max_correspondence_distance=1.0, max_iterations=ITERS_PER_ALIGN, rotation_eps=0.0, translation_eps=0.0, math_mode=args.math)
        regs = 30
        ref_pose = None
        for J in (1, 2, 3):
            ctxs = [sga.Context(0) for _ in range(J)]
            pbs = [sga.Problem(tree, src, np.eye(4), ctx=c) for c in ctxs]
            poses, errors = [None] * J, []
            for pb in pbs:
                for _ in range(2):
                    pb.align(st, np.eye(4))
            gate = threading.Barrier(J + 1)

            def work(j):
                try:
                    gate.wait()
                    for _ in range(regs):
                        poses[j] = pbs[j].align(st, np.eye(4)).T_target_source
                    ctxs[j].synchronize()
                except BaseException as ex:  # noqa: BLE001
                    errors.append(repr(ex))

            threads = [threading.Thread(target=work, args=(j,)) for j in range(J)]
            for th in threads:
                th.start()
            gate.wait()
            t0 = time.perf_counter()
            for th in threads:
                th.join()
            wall = time.perf_counter() - t0
            if errors:
                raise RuntimeError(errors[0])
            if ref_pose is None:
                ref_pose = poses[0]
            out["jobs"][str(J)] = {"iterations_per_s": J * regs * ITERS_PER_ALIGN / wall, "per_job_iterations_per_s": regs * ITERS_PER_ALIGN / wall,
                                   "poses_identical": bool(all(np.array_equal(ref_pose, q) for q in poses))}
            del pbs, ctxs
        out["best_iterations_per_s"] = max(v["iterations_per_s"] for v in out["jobs"].values())
        out["note"] = ("aggregate iterations/s of J independent registrations of the C3 pair on one GPU (%d registrations of %d iterations per job, one stream and "
                       "one host thread each, shared target index and source cloud); not the headline" % (regs, ITERS_PER_ALIGN))
    except Exception as ex:  # noqa: BLE001
        out["error"] = repr(ex)
    finally:
        sys.setswitchinterval(old_switch)
    return out


def scaled_scenes_leg(sga, ctx, target, source, args, headline):
    """The headline's steps on copies of the C3 clouds in other units of length (coordinates x0.01 and x10, the rejector's reach with
    them): the pass routing measures motions in units of the target's own length scale (csrc/linearize.hip: routing_unit), so the rate
    must not depend on the unit (within 3 % of C3).  LM's damping is not scale invariant (H_rr grows with the square of
    the coordinates), so the iterates — and with them the passes — differ a little from the metre-scale run's; tests/test_scale_free.py
    replays identical pose sequences and finds identical passes."""
    out = {}
    try:
        for scale in (0.01, 10.0):
            tgt = sga.PointCloud((target.astype(np.float64) * scale).astype(np.float32), ctx=ctx)
            src = sga.PointCloud((source.astype(np.float64) * scale).astype(np.float32), ctx=ctx)
            sga.estimate_covariances(tgt, None, args.neighbors)
            sga.estimate_covariances(src, None, args.neighbors)
            tree = sga.KdTree(tgt)
            problem = sga.Problem(tree, src, np.eye(4))
            st = sga.make_setting("GICP", max_correspondence_distance=1.0 * scale, max_iterations=ITERS_PER_ALIGN, rotation_eps=0.0, translation_eps=0.0, math_mode=args.math)
            for _ in range(2):
                problem.align(st, np.eye(4))
            s0 = problem.pass_stats()
            ctx.synchronize()
            t0 = time.perf_counter()
            steps = 0
            while steps < 200:
                steps += problem.align(st, np.eye(4)).iterations + 1
            ctx.synchronize()
            el = time.perf_counter() - t0
            s1 = problem.pass_stats()
            regs = steps / ITERS_PER_ALIGN
            out["x%g" % scale] = {"iterations_per_s": steps / el, "relative_to_headline": steps / el / headline if headline else None, "target_length_scale": tree.spacing(),
                                  "passes_per_registration": {"cold": (s1["cold_passes"] - s0["cold_passes"]) / regs, "warm": (s1["warm_passes"] - s0["warm_passes"]) / regs,
                                                              "walkers": (s1["walked_points"] - s0["walked_points"]) / regs}}
            del problem, tree, tgt, src
        return out
    except Exception as ex:  # noqa: BLE001
        out["error"] = repr(ex)
        return out


def preprocess_rooflines(sga, ctx, target_c3, args):
    """Roofline lines of the build-time stages (SURVEY 8a rows p1, p2 and the index build of a2): GPU time by HIP events
    (sga_debug_timer_*, the stage's launches and the gaps between them), algorithmic bytes, fraction of the HBM peak — at the C3 size
    (the 1M-point target of the headline) and at the C5 size (one KITTI-shaped scan: 115k raw points, ~11.5k after the 0.25 m grid).
    Algorithmic bytes: voxel grid 12 B read per input point + 12 B written per voxel; covariances (12 + 12 k) B read + 36 B written
    per point (the neighbours' coordinates are what the estimate consumes: normal_estimation.hpp:71-83); index build 12 B read per
    point + the bytes of the finished index (kd-ordered records, leaf blocks, nodes, pair records, boxes, group headers)."""
    try:
        k = args.neighbors
        out = {}

        def index_bytes(n):
            D = 0
            while ((n + (1 << D) - 1) >> D) > 8:
                D += 1
            return 12 * n + 16 * (n + 8) + (128 << D) // 1 + (8 << D) + (16 << D) + (32 << (D + 1)) + (128 << max(D - 2, 0))

        def line(alg_bytes, ms):
            us = 1e3 * ms
            gbs = alg_bytes / (us * 1e-6) / 1e9 if us > 0 else None
            return {"gpu_us": us, "algorithmic_bytes": int(alg_bytes), "achieved_GBs": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS if gbs else None}

        def stages(points, tag, leaf):
            n_raw = len(points)
            raw = sga.PointCloud(points, ctx=ctx)
            ctx.synchronize()
            best = {}
            for rep in range(4):  # the best of 4 (the first call of a size carries first-touch allocations)
                ms_vg, down = ctx.gpu_time_ms(lambda: sga.voxelgrid_sampling(raw, leaf))
                m = down.size()
                cloud = down if tag == "c5" else sga.PointCloud(points, ctx=ctx)  # C3: the index and the covariances are those of the 1M-point cloud itself
                n = cloud.size()
                ms_kd, tree = ctx.gpu_time_ms(lambda: sga.KdTree(cloud))
                ms_cov, _ = ctx.gpu_time_ms(lambda: sga.estimate_covariances(cloud, tree, k))
                cur = {"voxelgrid": line(12 * n_raw + 12 * m, ms_vg), "index_build": line(index_bytes(n), ms_kd), "covariances_k%d" % k: line((12 + 12 * k + 36) * n, ms_cov),
                       "points": {"voxelgrid_in": n_raw, "voxelgrid_out": m, "index_and_covariances": n, "voxel_size_m": leaf}}
                for name, v in cur.items():
                    if name == "points" or name not in best or v["gpu_us"] < best[name]["gpu_us"]:
                        best[name] = v
            return best

        out["c3_1M"] = stages(np.ascontiguousarray(target_c3[:, :3], dtype=np.float32), "c3", 0.25)
        out["c5_scan"] = stages(np.ascontiguousarray(sga.synthetic.kitti_like_scan(5)[0][:, :3], dtype=np.float32), "c5", 0.25)
        out["note"] = ("gpu_us = HIP events around the stage's launches (the gaps between them included); best of 4 calls; covariances of <= 32768 points search with one wave per query (csrc/knn_wave.hpp), "
                       "larger clouds with one query per lane; index build of <= 32768 points: one launch per level (radix select + partition), larger ones per-level segmented sorts")
        return out
    except Exception as ex:  # noqa: BLE001
        return {"error": repr(ex)}


def fresh_align_leg(sga, ctx, target, source, args):
    """What a caller of align() on two NEW clouds waits for (registration_helper.hpp:28-44: preprocess both clouds, register): from two
    host arrays (fp32, pageable) to the pose — upload, covariances k = 20 of both clouds, the target's kd-tree, the source sorted into
    the problem, GICP from the identity with the default termination criteria.  Wall time of the whole chain on the context in
    stream-ordered mode (one host wait, at the end) and, from a second run with a synchronisation after every stage, the stages.  Median of 5."""
    try:
        k = args.neighbors
        tp = np.ascontiguousarray(target[:, :3], dtype=np.float32)
        sp = np.ascontiguousarray(source[:, :3], dtype=np.float32)
        st = sga.make_setting("GICP", max_correspondence_distance=1.0)

        def chain(sync):
            t = [time.perf_counter()]

            def mark():
                if sync:
                    ctx.synchronize()
                    t.append(time.perf_counter())

            tgt, src = sga.PointCloud(tp, ctx=ctx), sga.PointCloud(sp, ctx=ctx)
            mark()
            tree = sga.KdTree(tgt)
            mark()
            sga.estimate_covariances(tgt, tree, k)
            sga.estimate_covariances(src, None, k)
            mark()
            pb = sga.Problem(tree, src, np.eye(4))
            mark()
            res = pb.align(st, np.eye(4))
            ctx.synchronize()
            t.append(time.perf_counter())
            return t, res

        prev = ctx.set_stream_ordered(True)
        try:
            chain(False)  # first-touch allocations of these sizes
            walls, stage_rows, res = [], [], None
            for _ in range(5):
                t, res = chain(False)
                walls.append(1e3 * (t[-1] - t[0]))
            for _ in range(3):
                t, _ = chain(True)
                stage_rows.append([1e3 * (b - a) for a, b in zip(t[:-1], t[1:])])
        finally:
            ctx.set_stream_ordered(prev)
        stages = np.median(np.array(stage_rows), axis=0)
        names = ["upload_2_clouds", "target_kdtree", "covariances_2_clouds", "problem_source_sort", "align_default_criteria"]
        return {"ms": float(np.median(walls)), "runs_ms": walls, "stages_ms_synchronised": {n: float(v) for n, v in zip(names, stages)}, "iterations": int(res.iterations) + 1, "converged": bool(res.converged),
                "points": [int(len(tp)), int(len(sp))],
                "note": "host arrays (fp32, pageable) -> pose: 2 uploads (2 x 12 MB: one staging pass on the host each, then PCIe), covariances k = %d of both clouds (the source's over a temporary tree of its own), the target's index, the source sorted by target leaf, "
                        "GICP from the identity with the default termination criteria; the clouds and the index are released between runs" % k}
    except Exception as ex:  # noqa: BLE001
        return {"error": repr(ex)}


def odometry_leg(sga, args, shard):
    """Second half of the BASELINE metric: ms/scan of scan-to-scan GICP odometry on the KITTI-shaped synthetic stream (C5), GPU vs
    the CPU oracle following the same protocol (downsample 0.25 m -> covariances k = 20 -> GICP against the previous scan).
    shard = (rank, world, ctx): BASELINE config 5 — every rank preprocesses the scan, registers its contiguous shard of the source and
    the accumulators are all-reduced once per linearization / error pass."""
    try:
        from small_gicp_amd import odometry

        if shard is not None:
            rank, world, ctx = shard
            r = odometry.run_synthetic(args.odom_frames, ctx=ctx, shard=(rank, world))
            out = {k: v for k, v in r.items() if k not in ("estimated", "ground_truth")}
            out["unit"] = "ms/scan"
            out["sharding"] = "source points of every scan split into %d contiguous shards, target index and preprocessing replicated, one all-reduce of 96 doubles per linearization (the system + the error-model moments)" % world
            return out
        import gc

        r = odometry.run_synthetic(args.odom_frames)
        out = {k: v for k, v in r.items() if k not in ("estimated", "ground_truth")}
        out["unit"] = "ms/scan"
        try:  # the same loop with the scans preloaded into PINNED host memory (sga_host_alloc): the upload has no CPU pass
            rp = odometry.run_synthetic(args.odom_frames, pinned=True)
            out["scans_in_pinned_host_memory"] = {"registration_ms_per_scan": rp["registration_ms_per_scan"], "total_ms_per_scan": rp["total_ms_per_scan"],
                                                  "poses_identical": bool(all(np.array_equal(a, b) for a, b in zip(rp["estimated"], r["estimated"])))}
        except Exception as ex:  # noqa: BLE001
            out["scans_in_pinned_host_memory"] = {"error": repr(ex)}
        try:
            best = None
            # the stages of the reference's flow graph (odometry_benchmark_small_gicp_tbb_flow.cpp:55-141) on HIP streams: preprocessing workers
            # beside registration workers (pairs are registered from the identity: independent); each run on contexts of its own, released before the next
            for workers, reg_workers in ((1, 1), (2, 1), (2, 2)):
                gc.collect()
                pr = odometry.run_synthetic_pipelined(max(args.odom_frames, 36), workers=workers, reg_workers=reg_workers)
                same = all(np.array_equal(a, b) for a, b in zip(pr["estimated"], r["estimated"]))
                out.setdefault("pipelined_by_workers", {})["%dx%d" % (workers, reg_workers)] = {"ms_per_scan": pr["ms_per_scan"], "poses_identical": bool(same)}
                if best is None or pr["ms_per_scan"] < best[0]:
                    best = (pr["ms_per_scan"], same, "%d preprocessing x %d registration (Python threads)" % (workers, reg_workers))
                del pr
            out["pipelined_total_ms_per_scan"] = best[0]  # throughput with the frames' chains running side by side (HIP streams)
            out["pipelined_poses_identical"] = bool(best[1])
            out["pipelined_workers"] = best[2]
        except Exception as ex:  # noqa: BLE001
            out["pipelined_error"] = repr(ex)
        gc.collect()
        try:  # the same flow as a C++ program (examples/odometry_benchmark_flow.cpp: std::thread workers, no interpreter between the calls)
            import tempfile, shutil

            work = tempfile.mkdtemp(prefix="sga_flow_")
            try:
                fl = {}
                for pw, rw in ((2, 1), (2, 2)):
                    fr = odometry.run_synthetic_cpp_flow(max(args.odom_frames, 36), workdir=work, preprocess_workers=pw, registration_workers=rw, repeat=4)
                    nf = min(len(fr["estimated"]), len(r["estimated"]))
                    fl["%dx%d" % (pw, rw)] = {"ms_per_scan": fr["ms_per_scan"], "runs_ms_per_scan": fr["runs_ms_per_scan"], "frame_latency_ms": fr["frame_latency_ms"], "mean_iterations": fr["mean_iterations"],
                                              "max_abs_pose_difference_vs_python_driver": float(max(np.abs(a - b).max() for a, b in zip(fr["estimated"][:nf], r["estimated"][:nf])))}
                bestk = min(fl, key=lambda k: fl[k]["ms_per_scan"])
                out["flow_cpp_driver"] = {"total_ms_per_scan": fl[bestk]["ms_per_scan"], "workers": bestk + " (preprocessing x registration)", "by_workers": fl, "driver": fr["driver"],
                                          "protocol": "src/benchmark/odometry_benchmark_small_gicp_tbb_flow.cpp:50-141 (total throughput = elapsed / frames, :113-114); trajectory written with 6 decimals"}
                if "pipelined_total_ms_per_scan" not in out or fl[bestk]["ms_per_scan"] < out["pipelined_total_ms_per_scan"]:
                    out["pipelined_total_ms_per_scan"] = fl[bestk]["ms_per_scan"]
                    out["pipelined_workers"] = bestk + " (C++ flow driver)"
                    out["pipelined_poses_identical"] = bool(fl[bestk]["max_abs_pose_difference_vs_python_driver"] < 1e-6)  # the text trajectory carries 6 decimals
            finally:
                shutil.rmtree(work, ignore_errors=True)
        except Exception as ex:  # noqa: BLE001
            out["flow_cpp_driver"] = {"error": repr(ex)}
        gc.collect()
        try:  # the reference's scan-to-MODEL engines (odometry_benchmark_small_vgicp_model_omp.cpp:12-57, ..._small_gicp_model_omp.cpp): one voxel map accumulating
            # every registered scan (incremental insert with the estimated pose, LRU removal), each scan registered against it from the previous pose
            frames_m = min(args.odom_frames, 60)
            sm = {}
            for model in ("gaussian", "flat"):
                mr = odometry.run_synthetic_model(frames_m, model=model)
                sm[model] = {k: mr[k] for k in ("registration_ms_per_scan", "total_ms_per_scan", "mean_iterations", "ate_trans_m_max", "num_voxels")}
            sm["frames"] = frames_m
            sm["protocol"] = ("gaussian: GaussianVoxelMap 1.0 m + VGICP (small_vgicp_model); flat: IncrementalVoxelMap<FlatContainerCov> + GICP (small_gicp_model); registration = covariances + align + "
                              "insert into the map; total adds the 0.25 m voxel grid; ate = largest drift of the trajectory against the generator's ground truth")
            out["scan_to_model"] = sm
        except Exception as ex:  # noqa: BLE001
            out["scan_to_model"] = {"error": repr(ex)}
        gc.collect()
        out["protocol"] = "src/benchmark/odometry_benchmark_small_gicp_omp.cpp:16-49: registration = index build + covariances + align; total adds the 0.25 m voxel grid"
        try:  # the same scans through the C++ driver (examples/odometry_benchmark.cpp): the reference's benchmark is a C++ program too
            cr = odometry.run_synthetic_cpp(args.odom_frames)
            out["cpp_driver"] = {"registration_ms_per_scan": cr["registration_ms_per_scan"], "total_ms_per_scan": cr["total_ms_per_scan"], "mean_iterations": cr["mean_iterations"],
                                 "max_abs_pose_difference_vs_python_driver": float(max(np.abs(a - b).max() for a, b in zip(cr["estimated"], r["estimated"]))), "driver": cr["driver"]}
        except Exception as ex:  # noqa: BLE001
            out["cpp_driver"] = {"error": repr(ex)}
        if not args.no_cpu_baseline:
            from oracle import orc, ref

            ncpu = os.cpu_count() or 1
            threads = max(1, min(32, ncpu))
            use_ref = ref.available()
            frames = min(args.odom_frames, 30)  # a bounded sample: ~3 ms per scan
            prev = None
            reg_ms = []
            for f in range(frames):
                pts, _ = sga.synthetic.kitti_like_scan(f)
                # protocol of odometry_benchmark_small_gicp_omp.cpp:16-49: the (already downsampled) scan -> KdTreeBuilderOMP ->
                # estimate_covariances_omp -> Registration<GICPFactor, ParallelReductionOMP>::align against the previous scan
                if use_ref:
                    down = ref.Cloud(pts.astype(np.float64), tree=False).voxelgrid_sampling(0.25).get()[0]
                    t0 = time.perf_counter()
                    cloud = ref.Cloud(down, tree=True, tree_threads=threads)
                    cloud.estimate_covariances(20, threads)
                    if prev is not None:
                        ref.align(prev, cloud, ref.GICP, 1.0, 1.0, threads)
                else:
                    down = orc.voxelgrid_sampling(pts, 0.25)
                    t0 = time.perf_counter()
                    cloud = orc.Cloud(down, tree=True)
                    cloud.estimate_normals_covariances(20, threads)
                    if prev is not None:
                        orc.align(prev, cloud, orc.default_setting(factor_kind=orc.GICP, num_threads=threads))
                reg_ms.append(1e3 * (time.perf_counter() - t0))
                prev = cloud
            out["cpu_registration_ms_per_scan"] = float(np.mean(reg_ms[1:])) if len(reg_ms) > 1 else None
            out["cpu_frames"] = frames
            out["cpu_threads"] = threads
            out["cpu_kind"] = ("reference (oracle/_ref: KdTreeBuilderOMP + estimate_covariances_omp + registration_helper.cpp align of the unmodified reference over the scalar Eigen stand-in, -O3)"
                               if use_ref else "port (oracle/ restatement, -O3 -fopenmp)")
        return out
    except Exception as ex:  # noqa: BLE001
        return {"error": repr(ex)}


def cpu_baseline(sga, tgt, src, n, args):
    """CPU baseline on the SAME clouds and covariances, timed region = the optimizer loop only (index build excluded, as on the GPU
    side).  kind "reference": the unmodified reference code (registration_helper.cpp align -> Registration<GICPFactor,
    ParallelReductionOMP>) from oracle/_ref, when that library travelled with the repository; kind "port": the oracle's restatement
    of the same path (oracle/), otherwise.  Protocol: one run per thread count {all, 1/2, 1/4, 1/8 of the host
    threads} to find the best count, then the MEDIAN of three runs at that count is the value; beside it the reference's default
    num_threads = 4 (registration_helper.hpp / reduction_omp.hpp:22) and the same sources compiled with -march=x86-64-v3 (the
    reference's BUILD_WITH_MARCH_NATIVE switch is off by default, CMakeLists.txt:30).
    Returns (json object, CPU result or None, ref_nn or None)."""
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

        def clouds():
            t0 = time.perf_counter()
            if use_ref:
                a, b = ref.Cloud(tp, None, tcov, tree=True, tree_threads=min(32, ncpu)), ref.Cloud(sp, None, scov, tree=False)
            else:
                a, b = orc.Cloud(tp, None, tcov, tree=True), orc.Cloud(sp, None, scov, tree=False)
            return a, b, time.perf_counter() - t0

        def run(otc, osc, threads, iters):
            if use_ref:
                return ref.align(otc, osc, ref.GICP, 1.0, 1.0, threads, iters, 0.0, 0.0)
            s = orc.default_setting(factor_kind=orc.GICP, num_threads=threads, max_iterations=iters, rotation_eps=0.0, translation_eps=0.0)
            return orc.align(otc, osc, s)

        otc, osc, build_s = clouds()
        tried = {}
        best = None
        for threads in counts:
            r = run(otc, osc, threads, args.cpu_iters)
            ips = (r.iterations + 1) / r.elapsed_sec
            tried[str(threads)] = ips
            if best is None or ips > best[0]:
                best = (ips, threads)
        threads = best[1]
        runs = [run(otc, osc, threads, args.cpu_iters) for _ in range(3)]
        rates = sorted((r.iterations + 1) / r.elapsed_sec for r in runs)
        r = runs[0]
        ips = rates[1]
        # the same run with the threads pinned: libgomp reads OMP_PROC_BIND / OMP_PLACES when it is loaded, so each
        # placement gets a process of its own (scripts/cpu_ref_rate.py) on the same clouds
        bound = {}
        try:
            import tempfile

            with tempfile.TemporaryDirectory(dir="/tmp") as td:
                npz = os.path.join(td, "clouds.npz")
                np.savez(npz, tp=tp, tcov=tcov, sp=sp, scov=scov)
                for bind, places, th in (("close", "cores", threads), ("spread", "cores", threads), ("close", "cores", min(ncpu, 2 * threads))):
                    env = dict(os.environ, OMP_PROC_BIND=bind, OMP_PLACES=places)
                    pr = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "cpu_ref_rate.py"), npz, str(th), str(args.cpu_iters)], capture_output=True, text=True, timeout=600, env=env)
                    key = "OMP_PROC_BIND=%s OMP_PLACES=%s threads=%d" % (bind, places, th)
                    bound[key] = json.loads(pr.stdout.strip().splitlines()[-1])["iterations_per_s"] if pr.returncode == 0 and pr.stdout.strip() else "failed: " + pr.stderr[-200:]
        except Exception as ex:  # noqa: BLE001
            bound["error"] = repr(ex)
        best_bound = max([v for v in bound.values() if isinstance(v, float)], default=0.0)
        unbound_median = ips
        if best_bound > ips:  # report the best
            ips = best_bound
        few = max(2, min(args.cpu_iters, 3))
        r4 = run(otc, osc, min(4, ncpu), few)
        default_threads = {"threads": min(4, ncpu), "iterations_per_s": (r4.iterations + 1) / r4.elapsed_sec, "lm_iterations_timed": r4.iterations + 1}
        v3 = None
        ref_nn = None
        if use_ref:
            nn_tree = otc

            def ref_nn(T):  # the reference's own search + DistanceRejector at pose T, caller's order
                q = sp @ np.asarray(T)[:3, :3].T + np.asarray(T)[:3, 3]
                idx, d2 = nn_tree.nearest(q, min(32, ncpu))
                return np.where(d2 > 1.0, -1, idx)

            if os.path.exists(ref.LIB_PATH_V3):
                try:
                    default_path = ref.LIB_PATH
                    import ctypes as C

                    L3 = C.CDLL(ref.LIB_PATH_V3)  # a second copy of the same code: its own handles
                    v3 = _v3_rate(L3, tp, tcov, sp, scov, threads, args.cpu_iters, min(32, ncpu))
                    assert ref.LIB_PATH == default_path
                except Exception as ex:  # noqa: BLE001
                    v3 = {"error": repr(ex)}
        what = ("unmodified reference code (oracle/_ref: registration_helper.cpp align, Registration<GICPFactor, ParallelReductionOMP>) compiled -O3 without -march=native over a scalar, "
                "un-vectorised Eigen stand-in (oracle/ref/eigen_shim): the reference's algorithm and memory behaviour, not Eigen's SIMD kernels — a stated baseline, probably slower than a build against real Eigen"
                ) if use_ref else "oracle/ restatement of ParallelReductionOMP + KdTree + GICPFactor + LM"
        return {
            "value": ips,
            "unit": "iterations/s",
            "cores": threads,
            "kind": "reference" if use_ref else "port",
            "sample": "%s; full C3 pair (%d<->%d), %d outer LM iterations from identity, OpenMP schedule(guided,8); kd-tree build %.2fs excluded; value = the best of {median of 3 unpinned runs at the best thread count %s, "
                      "the pinned runs %s}; one unpinned run per thread count: %s.  Thread scaling turns NEGATIVE past ~32 threads on this host: the loop is a pointer-chasing kd descent per source point "
                      "(ann/kdtree.hpp:207-230) over a 1M-node tree plus 464 B of AoS reads and a 144 B factor write per point (point_cloud.hpp:69-71, gicp_factor.hpp:94-96): bound by memory latency and "
                      "by the guided,8 scheduler's shared counter, not by arithmetic — more threads add contention, and the scalar Eigen stand-in gives -march flags nothing to vectorise"
                      % (what, n, n, r.iterations + 1, build_s, json.dumps([round(x, 3) for x in rates]), json.dumps(bound), json.dumps(tried)),
            "unpinned_median_at_best_thread_count": unbound_median,
            "pinned_runs": bound,
            "runs_at_best_thread_count": rates,
            "by_thread_count": tried,
            "reference_default_num_threads_4": default_threads,
            "march_x86_64_v3": v3,
            "host_threads_available": ncpu,
        }, r, ref_nn
    except Exception as ex:  # noqa: BLE001
        return {"value": None, "unit": "iterations/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (ex,)}, None, None


def _v3_rate(L, tp, tcov, sp, scov, threads, iters, tree_threads):
    """One timed run of ref_align from the -march=x86-64-v3 build of oracle/_ref (its own ctypes handle: both builds stay loaded)."""
    import ctypes as C

    from oracle import ref

    vp, dp = C.c_void_p, C.POINTER(C.c_double)
    L.ref_cloud_create.argtypes = [dp, dp, dp, C.c_size_t, C.c_int, C.c_int]
    L.ref_cloud_create.restype = vp
    L.ref_cloud_destroy.argtypes = [vp]
    L.ref_align.argtypes = [vp, vp, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, C.c_double, dp, C.POINTER(ref.Result), dp]

    def arr(a, last):
        return np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1, last))

    a, ac, b, bc = arr(tp, 3), arr(np.asarray(tcov).reshape(len(tp), 9), 9), arr(sp, 3), arr(np.asarray(scov).reshape(len(sp), 9), 9)
    th = L.ref_cloud_create(a.ctypes.data_as(dp), None, ac.ctypes.data_as(dp), len(a), 1, tree_threads)
    sh = L.ref_cloud_create(b.ctypes.data_as(dp), None, bc.ctypes.data_as(dp), len(b), 0, 1)
    try:
        res, el = ref.Result(), C.c_double()
        t16 = np.ascontiguousarray(np.eye(4)).reshape(16)
        rc = L.ref_align(th, sh, ref.GICP, 1.0, 1.0, threads, iters, 0.0, 0.0, t16.ctypes.data_as(dp), C.byref(res), C.byref(el))
        assert rc == 0
        return {"threads": threads, "iterations_per_s": (int(res.iterations) + 1) / el.value, "flags": "-O3 -march=x86-64-v3 (AVX2 + FMA; -march=native of the build host would not be portable)"}
    finally:
        L.ref_cloud_destroy(th)
        L.ref_cloud_destroy(sh)


if __name__ == "__main__":
    main()
