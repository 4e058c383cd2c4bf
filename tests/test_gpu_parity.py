"""Parity of the HIP path against the CPU oracle — everything here runs on a real MI355X through the C-ABI.

Tolerances (written once, used below):
  * pose: 1e-4 m / 1e-4 rad against the oracle (BASELINE.json north_star), with identical iteration and inlier counts;
  * accumulators at a fixed pose on identical fp32 inputs: fp32 math 2e-5 relative to max|H| (resp. |e|), fp64 math 1e-10;
  * indices (NN / kNN / correspondences): equal, except where the two nearest candidates tie within fp32 resolution.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import small_gicp_amd as sga
from conftest import ROOT, pose_error

pytestmark = pytest.mark.gpu

POSE_TOL_T, POSE_TOL_R = 1e-4, 1e-4
FP32_REL, FP64_REL = 2e-5, 1e-10
C1_DIFFERING_FP32, C1_INLIER_DELTA_FP32 = 1, 1  # C1 at a fixed pose, fp32 arithmetic: correspondences (of 6 167) / inliers that may differ from the oracle's; observed on the MI355X: 0 / 0 for every factor and pose (gpurun_out/c1_observed.txt, round 6) — the 1 is for a tie within fp32 resolution


def rot(axis, ang):
    from scipy.spatial.transform import Rotation

    return Rotation.from_rotvec(np.asarray(axis, dtype=np.float64) / np.linalg.norm(axis) * ang).as_matrix()


def se3(axis, ang, t):
    T = np.eye(4)
    T[:3, :3] = rot(axis, ang)
    T[:3, 3] = t
    return T


@pytest.fixture(scope="module")
def gpu_c1(c1_f32):
    d = c1_f32
    tgt = sga.PointCloud(d["tp"], d["tn"], d["tc"])
    src = sga.PointCloud(d["sp"], d["sn"], d["sc"])
    tree = sga.KdTree(tgt)
    return tgt, src, tree


POSES = [np.eye(4), se3([0.1, 0.2, 1.0], np.deg2rad(0.7), [0.49, 0.12, -0.02]), se3([1, -1, 0.3], np.deg2rad(5.0), [-0.4, 0.3, 0.2])]
FACTORS = [("GICP", 2, None), ("PLANE_ICP", 1, None), ("ICP", 0, None), ("GICP", 2, "HUBER"), ("GICP", 2, "CAUCHY"), ("ICP", 0, "HUBER")]


@pytest.mark.parametrize("name,kind,robust", FACTORS)
@pytest.mark.parametrize("mode", ["fp32", "fp64"])
def test_linearize_and_error_match_oracle(orc, c1_f32, gpu_c1, name, kind, robust, mode):
    tgt, src, tree = gpu_c1
    otc, osc = c1_f32["otc"], c1_f32["osc"]
    rel = FP32_REL if mode == "fp32" else FP64_REL
    rk = {None: 0, "HUBER": 1, "CAUCHY": 2}[robust]
    st = sga.make_setting(name, robust_kernel=robust, robust_c=0.7, math_mode=mode)
    os_ = orc.default_setting(factor_kind=kind, robust_kind=rk, robust_c=0.7, num_threads=4)
    pb = sga.Problem(tree, src)
    f = orc.Factors(len(osc))
    for T in POSES:
        H, b, e, n = pb.linearize(st.factor, T)
        Ho, bo, eo, no = orc.linearize(otc, osc, os_, T, f)
        assert abs(int(n) - int(no)) <= (C1_INLIER_DELTA_FP32 if mode == "fp32" else 0)
        scale = np.abs(Ho).max()
        assert np.abs(H - Ho).max() <= rel * scale, (name, mode, np.abs(H - Ho).max() / scale)
        assert np.abs(H - H.T).max() == 0.0
        # b is a difference of large terms near the optimum: compare on the scale of |J^T M| ~ sqrt(H_ii * e)
        bscale = np.sqrt(np.abs(np.diag(Ho)) * max(eo, 1e-30)) + 1e-30
        assert (np.abs(b - bo) / bscale).max() <= 5 * rel, (name, mode, (np.abs(b - bo) / bscale).max())
        assert abs(e - eo) <= rel * abs(eo)
        # the error pass at the linearization point reproduces e (same correspondences, same cached mahalanobis)
        T2 = T @ se3([0, 0, 1], 1e-3, [1e-3, -2e-3, 5e-4])
        for Tq in (T, T2):
            assert abs(pb.error(st.factor, Tq) - orc.error(otc, osc, os_, Tq, f)) <= rel * abs(eo)
        ti, m6 = pb.factors()
        oti, om = f.get()
        differing = int((ti != oti).sum())
        print("C1 %s %s%s pose %d: inlier delta %d, correspondences differing %d of %d" % (name, mode, "/" + robust if robust else "", [np.array_equal(T, P) for P in POSES].index(True), int(n) - int(no), differing, len(ti)))
        assert differing <= (C1_DIFFERING_FP32 if mode == "fp32" else 0), differing
        if name == "GICP" and mode == "fp32":
            same = ti == oti
            om6 = np.stack([om[:, 0, 0], om[:, 0, 1], om[:, 0, 2], om[:, 1, 1], om[:, 1, 2], om[:, 2, 2]], axis=1)
            ok = same & (oti >= 0)
            assert np.abs(m6[ok] - om6[ok]).max() <= 1e-4 * np.abs(om6[ok]).max()


@pytest.mark.parametrize("name", ["GICP", "PLANE_ICP", "ICP"])
@pytest.mark.parametrize("mode", ["fp32", "fp64"])  # sga_linearize builds the error model in both arithmetic modes
def test_error_model_matches_error_kernel(gpu_c1, name, mode):
    """sga_error answers from the quadratic error model accumulated by the last linearization; the error kernel (a pass over the
    cloud with the cached correspondences / mahalanobis, Reduction::error of reduction_omp.hpp:61-70) must give the same number at
    every trial pose, small or large."""
    tgt, src, tree = gpu_c1
    st = sga.make_setting(name, math_mode=mode)
    pb = sga.Problem(tree, src)
    rel = 1e-5 if mode == "fp32" else 1e-10
    try:
        for T in POSES:
            _, _, e0, _ = pb.linearize(st.factor, T)
            trials = [T, T @ se3([0, 0, 1], 1e-3, [1e-3, -2e-3, 5e-4]), T @ se3([1, 2, -1], 0.05, [0.3, -0.2, 0.1]), se3([0.3, 1, 0.2], 0.4, [1.0, 2.0, -0.5]) @ T]
            sga.set_error_model(True)
            em = [pb.error(st.factor, Tq) for Tq in trials]
            sga.set_error_model(False)
            ek = [pb.error(st.factor, Tq) for Tq in trials]
            sga.set_error_model(True)
            assert abs(em[0] - e0) <= rel * abs(e0)
            for a, b in zip(em, ek):
                assert abs(a - b) <= rel * max(abs(b), abs(e0)), (name, mode, a, b)
    finally:
        sga.set_error_model(True)


@pytest.mark.parametrize("case", ["GICP", "PLANE_ICP", "ICP", "HUBER_GICP", "CAUCHY_GICP"])
@pytest.mark.parametrize("mode", ["fp32", "fp64"])
def test_align_matches_golden(c1_gold, gpu_c1, case, mode):
    """data/source.ply <-> data/target.ply (config C1): final pose within 1e-4 m / 1e-4 rad of the oracle."""
    tgt, src, tree = gpu_c1
    g = c1_gold["cases"][case]
    reg = case.split("_")[-1] if case not in ("PLANE_ICP",) else "PLANE_ICP"
    robust = case.split("_")[0] if case.startswith(("HUBER", "CAUCHY")) else None
    st = sga.make_setting(reg, robust_kernel=robust, math_mode=mode)
    res = sga.Problem(tree, src).align(st)
    dt, dr = pose_error(res.T_target_source, np.array(g["T"]))
    assert dt < POSE_TOL_T and dr < POSE_TOL_R, (case, mode, dt, dr)
    assert res.iterations == g["iterations"] and res.converged == g["converged"]
    assert abs(res.num_inliers - g["num_inliers"]) <= 2
    assert abs(res.error - g["error"]) <= 1e-4 * abs(g["error"])
    assert np.abs(res.H - np.array(g["H"])).max() <= 1e-3 * np.abs(np.array(g["H"])).max()


def test_vgicp_matches_golden(orc, c1_gold, c1_f32, gpu_c1):
    tgt, src, tree = gpu_c1
    g = c1_gold["cases"]["VGICP"]
    vm = sga.GaussianVoxelMap(1.0)
    vm.insert(tgt)
    assert vm.size() == g["num_voxels"]
    # voxel contents and ids (first-insertion order) against the oracle
    ovm = orc.VoxelMap(c1_f32["otc"], 1.0)
    oc, om, ocov, ocnt = ovm.get()
    gc, gm, gc6, gcnt = vm.download()
    assert (gc == oc).all() and (gcnt == ocnt).all()
    assert np.abs(gm - om).max() < 1e-5
    res = sga.align(vm, src)
    dt, dr = pose_error(res.T_target_source, np.array(g["T"]))
    assert dt < POSE_TOL_T and dr < POSE_TOL_R, (dt, dr)
    assert res.iterations == g["iterations"] and abs(res.num_inliers - g["num_inliers"]) <= 2


def test_full_gpu_pipeline_from_raw_points(c1_raw, c1_gold):
    """Config C1 end to end on the GPU: raw points -> voxel grid -> covariances -> index -> GICP, i.e. small_gicp::align(points...)."""
    tgt, src, T_gt = c1_raw
    res = sga.align(tgt, src, downsampling_resolution=0.25)
    dt, dr = pose_error(res.T_target_source, np.array(c1_gold["cases"]["GICP"]["T"]))
    assert dt < POSE_TOL_T and dr < POSE_TOL_R, (dt, dr)
    dt, dr = pose_error(res.T_target_source, T_gt)
    assert dt < 0.05 and dr < 0.05  # the reference's own pin (src/test/python_test.py:52-58)
    resv = sga.align(tgt, src, registration_type="VGICP", voxel_resolution=1.0)
    dt, dr = pose_error(resv.T_target_source, np.array(c1_gold["cases"]["VGICP"]["T"]))
    assert dt < POSE_TOL_T and dr < POSE_TOL_R, (dt, dr)


def test_init_guess_and_inverse(orc, c1_f32, gpu_c1, c1_raw):
    tgt, src, tree = gpu_c1
    otc, osc = c1_f32["otc"], c1_f32["osc"]
    T_gt = c1_raw[2]
    st = sga.make_setting("GICP")
    os_ = orc.default_setting(factor_kind=orc.GICP, num_threads=4)
    rng = np.random.default_rng(11)
    for _ in range(3):
        init = T_gt @ se3(rng.normal(size=3), np.deg2rad(rng.uniform(0, 8)), rng.uniform(-0.4, 0.4, 3))
        res = sga.Problem(tree, src, init).align(st, init)
        ref = orc.align(otc, osc, os_, init_T=init)
        dt, dr = pose_error(res.T_target_source, ref.T_target_source)
        assert dt < POSE_TOL_T and dr < POSE_TOL_R and res.iterations == ref.iterations
    # inverse direction: roles swapped
    osc_t = orc.Cloud(*[a for a in c1_f32["osc"].get()])
    otc_s = orc.Cloud(*[a for a in c1_f32["otc"].get()], tree=False)
    ref = orc.align(osc_t, otc_s, os_)
    res = sga.Problem(sga.KdTree(src), tgt).align(st)
    dt, dr = pose_error(res.T_target_source, ref.T_target_source)
    assert dt < POSE_TOL_T and dr < POSE_TOL_R
    dt, dr = pose_error(res.T_target_source, np.linalg.inv(T_gt))
    assert dt < 0.2 and dr < np.deg2rad(2.5)


def test_gauss_newton_and_null_rejector(orc, c1_f32, gpu_c1):
    tgt, src, tree = gpu_c1
    otc, osc = c1_f32["otc"], c1_f32["osc"]
    res = sga.Problem(tree, src).align(sga.make_setting("GICP", optimizer="GN"))
    ref = orc.align(otc, osc, orc.default_setting(factor_kind=orc.GICP, num_threads=4, optimizer_type=1))
    dt, dr = pose_error(res.T_target_source, ref.T_target_source)
    assert dt < POSE_TOL_T and dr < POSE_TOL_R and res.iterations == ref.iterations
    # NullRejector (rejector.hpp:11-16): every source point keeps its (unbounded) nearest neighbour
    st = sga.make_setting("ICP", max_correspondence_distance=None)
    pb = sga.Problem(tree, src)
    H, b, e, n = pb.linearize(st.factor, np.eye(4))
    f = orc.Factors(len(osc))
    Ho, bo, eo, no = orc.linearize(otc, osc, orc.default_setting(factor_kind=orc.ICP, max_dist_sq=1e300, num_threads=4), np.eye(4), f)
    assert n == no == src.size() and abs(e - eo) <= FP32_REL * eo


@pytest.mark.parametrize("optimizer", ["LM", "GN"])
@pytest.mark.parametrize("mode", ["fp32", "fp64"])
def test_restrict_dof_factor_matches_oracle(orc, c1_f32, gpu_c1, optimizer, mode):
    """RestrictDoFFactor (factors/general_factor.hpp:41-75: H += lambda |mask - 1|, update_error a no-op) under both optimizers:
    pose within 1e-4 m / 1e-4 rad of the oracle — which equals the reference's Registration<GICPFactor, ParallelReductionOMP,
    RestrictDoFFactor, ...> to 1e-9 (tests/test_oracle_vs_reference.py) — same iteration and inlier counts, and the frozen
    directions (roll, pitch, z) stay frozen."""
    tgt, src, tree = gpu_c1
    mask = (0.0, 0.0, 1.0, 1.0, 1.0, 0.0)
    res = sga.Problem(tree, src).align(sga.make_setting("GICP", optimizer=optimizer, math_mode=mode, restrict_dof_lambda=1e9, restrict_dof_mask=mask))
    ref = orc.align(c1_f32["otc"], c1_f32["osc"], orc.default_setting(factor_kind=orc.GICP, num_threads=4, optimizer_type=0 if optimizer == "LM" else 1, restrict_lambda=1e9, restrict_mask=mask))
    dt, dr = pose_error(res.T_target_source, ref.T_target_source)
    assert dt < POSE_TOL_T and dr < POSE_TOL_R and res.iterations == ref.iterations and res.converged == ref.converged, (dt, dr, res.iterations, ref.iterations)
    assert abs(int(res.num_inliers) - int(ref.num_inliers)) <= 2
    T = res.T_target_source
    assert abs(T[2, 3]) < 1e-3 and abs(T[2, 0]) < 1e-3 and abs(T[2, 1]) < 1e-3
    # and it differs from the unrestricted optimum (the constraint is active on this pair: ground truth has z = -0.025)
    free = sga.Problem(tree, src).align(sga.make_setting("GICP", optimizer=optimizer, math_mode=mode))
    assert abs(free.T_target_source[2, 3]) > 5e-3


def test_host_rejector_callback(gpu_c1):
    """A user-supplied CorrespondenceRejector (rejector.hpp:11-28 is a duck-typed functor) through sga_problem_set_rejector: a callback
    equivalent to DistanceRejector reproduces the built-in one exactly, one the built-in cannot express (a predicate on the target
    index) filters exactly the pairs it names, and None restores the built-in behaviour."""
    tgt, src, tree = gpu_c1
    st = sga.make_setting("GICP", max_correspondence_distance=0.5)
    st_other = sga.make_setting("GICP", max_correspondence_distance=7.0)  # ignored while a callback is installed
    base = sga.Problem(tree, src).linearize(st.factor, np.eye(4))
    pb = sga.Problem(tree, src)
    seen = []

    def like_distance_rejector(T, target_index, sq_dist):
        seen.append((len(target_index), int((target_index >= 0).sum())))
        return sq_dist > 0.25

    pb.set_rejector(like_distance_rejector)
    got = pb.linearize(st_other.factor, np.eye(4))
    assert seen and seen[0] == (src.size(), src.size())  # the callback saw every source point with its (unbounded) nearest neighbour
    # (the callback path runs the separate factor kernel: other fma contractions per point, another order of the fp64 sums)
    assert got[3] == base[3] and np.abs(got[0] - base[0]).max() <= 1e-7 * np.abs(base[0]).max() and abs(got[2] - base[2]) <= 1e-7 * abs(base[2])
    r_cb = pb.align(st_other)
    r_builtin = sga.Problem(tree, src).align(st)
    assert r_cb.iterations == r_builtin.iterations and np.abs(r_cb.T_target_source - r_builtin.T_target_source).max() < 1e-6
    pb.set_rejector(lambda T, target_index, sq_dist: (target_index % 2 == 1) | (sq_dist > 1.0))
    H, b, e, n = pb.linearize(st.factor, np.eye(4))
    corr = pb.factors()[0]
    assert n == (corr >= 0).sum() > 1000 and (corr[corr >= 0] % 2 == 0).all()
    pb.set_rejector(None)
    again = pb.linearize(st.factor, np.eye(4))
    assert again[3] == base[3] and abs(again[2] - base[2]) <= 1e-7 * abs(base[2])


# ---- nearest-neighbour search (kdtree_test.cpp / kdtree_synthetic_test.cpp protocols) ----------------------------------------
def _brute(target, queries, k):
    d2 = ((queries[:, None, :].astype(np.float64) - target[None, :, :].astype(np.float64)) ** 2).sum(-1)
    idx = np.argsort(d2, axis=1, kind="stable")[:, :k]
    return idx, np.take_along_axis(d2, idx, axis=1)


def test_knn_real_data(c1_f32, gpu_c1):
    tgt, src, tree = gpu_c1
    pts = c1_f32["tp"]
    rng = np.random.default_rng(1)
    q = np.concatenate([pts[rng.choice(len(pts), 50, replace=False)], pts[rng.choice(len(pts), 50, replace=False)] + rng.normal(0, 1.0, (50, 3)).astype(np.float32), rng.uniform(0, 100, (50, 3)).astype(np.float32)]).astype(np.float32)
    idx, d2 = tree.batch_knn_search(q, 20)
    bi, bd = _brute(pts, q, 20)
    assert (np.abs(d2 - bd) <= 1e-3 * np.minimum(1.0, bd) + 3e-7 * bd).all()  # kdtree_test.cpp:81-105 (1e-3), fp32 resolution for the far queries
    assert (idx == bi).mean() > 0.999  # fp32 near-ties may swap neighbours
    i1, d1 = tree.batch_nearest_neighbor_search(q)
    assert (i1 == bi[:, 0]).mean() > 0.995 and (np.abs(d1 - bd[:, 0]) <= 1e-3 * np.minimum(1.0, bd[:, 0]) + 3e-7 * bd[:, 0]).all()
    # bounded search: everything farther than the radius is reported as not found
    ib, db = tree.batch_knn_search(q, 5, max_sq_dist=0.25)
    assert ((ib >= 0) == (bd[:, :5] <= 0.25 + 1e-6)).mean() > 0.999


SYN = {
    "uniform1": lambda rng: rng.uniform(-1, 1, (256, 3)),
    "uniform1e6": lambda rng: rng.uniform(-1e6, 1e6, (256, 3)),
    "bimodal": lambda rng: np.concatenate([rng.normal(-5, 0.5, (128, 3)), rng.normal(5, 0.5, (128, 3))]),
    "lattice": lambda rng: rng.integers(-3, 4, (256, 3)).astype(np.float64),
}


@pytest.mark.parametrize("name", list(SYN))
@pytest.mark.parametrize("ntrunc", [256, 10, 5])
def test_knn_synthetic(name, ntrunc):
    rng = np.random.default_rng(3)
    target = SYN[name](rng)[:ntrunc].astype(np.float32)
    queries = SYN[name](rng).astype(np.float32)
    tree = sga.KdTree(sga.PointCloud(target))
    idx, d2 = tree.batch_knn_search(queries, 20)
    k = min(20, ntrunc)
    _, bd = _brute(target, queries, k)
    scale = max(1.0, float(bd.max()))
    assert np.abs(d2[:, :k] - bd).max() <= 2e-6 * scale  # fp32 distances
    assert (idx[:, k:] == -1).all() and (idx[:, :k] >= 0).all()
    # returned distance is consistent with the returned index
    chk = ((queries[:, None, :].astype(np.float64) - target[idx[:, :k]].astype(np.float64)) ** 2).sum(-1)
    assert np.abs(chk - d2[:, :k]).max() <= 2e-6 * scale


def test_empty_and_tiny_inputs(capfd):
    empty = sga.PointCloud(np.zeros((0, 3), np.float32))
    assert empty.size() == 0 and sga.voxelgrid_sampling(empty, 0.5).size() == 0  # downsampling.hpp:24-26
    tree = sga.KdTree(empty)
    idx, d2 = tree.batch_knn_search(np.zeros((4, 3), np.float32), 3)
    assert (idx == -1).all()
    pts = np.random.default_rng(0).uniform(-1, 1, (8, 3)).astype(np.float32)
    cov = np.tile(np.eye(3, dtype=np.float32), (8, 1, 1))
    small = sga.PointCloud(pts, covs=cov)
    # empty source against a real target and vice versa: zero system, no crash, warning like registration.hpp:34-39
    st = sga.make_setting("GICP")
    pb = sga.Problem(sga.KdTree(small), sga.PointCloud(np.zeros((0, 3), np.float32), covs=np.zeros((0, 3, 3), np.float32)))
    H, b, e, n = pb.linearize(st.factor, np.eye(4))
    assert n == 0 and e == 0 and not H.any()
    pb2 = sga.Problem(sga.KdTree(sga.PointCloud(np.zeros((0, 3), np.float32), covs=np.zeros((0, 3, 3), np.float32))), small)
    H, b, e, n = pb2.linearize(st.factor, np.eye(4))
    assert n == 0 and e == 0
    res = pb2.align(st)
    assert not res.converged or res.num_inliers == 0
    assert "too small" in capfd.readouterr().err
    # a source far away from the target: every pair is rejected
    far = sga.PointCloud(pts + 100.0, covs=cov)
    H, b, e, n = sga.Problem(sga.KdTree(small), far).linearize(st.factor, np.eye(4))
    assert n == 0 and e == 0


def test_error_reporting():
    pts = np.random.default_rng(0).uniform(-1, 1, (64, 3)).astype(np.float32)
    nocov = sga.PointCloud(pts)
    with pytest.raises(sga.SgaError):
        sga.Problem(sga.KdTree(nocov), nocov).linearize(sga.make_setting("GICP").factor, np.eye(4))
    with pytest.raises(sga.SgaError):
        sga.Problem(sga.KdTree(nocov), nocov).linearize(sga.make_setting("PLANE_ICP").factor, np.eye(4))
    with pytest.raises(sga.SgaError):
        vm = sga.GaussianVoxelMap(1.0)
        vm.insert(nocov)


# ---- preprocessing ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("leaf", [0.1, 0.25, 1.0])
def test_voxelgrid_matches_oracle(orc, c1_raw, leaf):
    pts = c1_raw[0]
    out = sga.voxelgrid_sampling(pts, leaf).xyz()
    ref = orc.voxelgrid_sampling(pts, leaf)
    assert out.shape == ref.shape  # same voxels, same (ascending key) order
    assert np.abs(out - ref).max() < 1e-5


def _neighbourhood_differences(gn, gcov, on, ocov):
    """Per-point comparison of normals / covariances: the points whose neighbourhood (or whose near-degenerate spectrum) came out
    different on the two sides — an entry off by more than 1e-3; everything else must agree to 1e-5."""
    dn = np.abs(gn - on).max(axis=1)
    dc = np.abs(gcov - ocov).reshape(len(on), -1).max(axis=1)
    bad = (dn > 1e-3) | (dc > 1e-3)
    return bad, float(dn[~bad].max(initial=0.0)), float(dc[~bad].max(initial=0.0))


@pytest.mark.parametrize("k", [10, 20])
def test_normals_covariances_match_oracle(orc, c1_raw, k):
    """util/normal_estimation.hpp:65-92 on C1 (6k points): every point but a counted handful — neighbourhoods whose k-th and (k+1)-th
    neighbours tie within fp32 resolution (the device searches fp32 distances, the CPU doubles), or whose two smallest eigenvalues
    nearly coincide — equals the oracle to 1e-5."""
    down = orc.voxelgrid_sampling(c1_raw[0], 0.25).astype(np.float32)
    cloud = sga.PointCloud(down)
    sga.estimate_normals_covariances(cloud, None, k)
    oc = orc.Cloud(down.astype(np.float64))
    oc.estimate_normals_covariances(k, 4)
    _, on, ocov = oc.get()
    gn, gcov = cloud.normals()[:, :3], cloud.covs()[:, :3, :3]
    assert np.abs(np.linalg.norm(gn, axis=1) - 1).max() < 1e-5  # normal_estimation_test.cpp: unit normals
    bad, worst_n, worst_c = _neighbourhood_differences(gn, gcov, on, ocov)
    print("C1 k=%d: %d of %d neighbourhoods differ; the others agree to %.1e (normals) / %.1e (covariances)" % (k, int(bad.sum()), len(on), worst_n, worst_c))
    assert int(bad.sum()) <= P2_MAX_DIFFERING_C1 and worst_n < 1e-5 and worst_c < 1e-5, (int(bad.sum()), worst_n, worst_c)
    assert np.abs(gcov - np.transpose(gcov, (0, 2, 1))).max() == 0
    # flows through an explicit index too, and updates the index's own attribute copies
    cloud2 = sga.PointCloud(down)
    tree = sga.KdTree(cloud2)
    sga.estimate_covariances(cloud2, tree, k)
    assert np.abs(cloud2.covs()[:, :3, :3] - gcov).max() < 1e-6
    few = sga.PointCloud(down[:4])
    sga.estimate_normals_covariances(few, None, k)  # < 5 neighbours: normal 0, cov I (normal_estimation.hpp:15,33-37)
    assert not few.normals()[:, :3].any() and np.allclose(few.covs()[:, :3, :3], np.eye(3))


# ---- size-independent properties at the benchmark size (config C3: 1M <-> 1M) ------------------------------------------------------
@pytest.fixture(scope="module")
def c3():
    target, source, T_gt = sga.synthetic.registration_pair(1_000_000)
    tgt, src = sga.PointCloud(target), sga.PointCloud(source)
    sga.estimate_covariances(tgt, None, 20)
    sga.estimate_covariances(src, None, 20)
    tree = sga.KdTree(tgt)
    return dict(tgt=tgt, src=src, tree=tree, T_gt=T_gt, source=source, src_cov=src.covs()[:, :3, :3].astype(np.float32))


def test_c3_properties(c3):
    st = sga.make_setting("GICP")
    pb = sga.Problem(c3["tree"], c3["src"])
    T = se3([0.2, 0.3, 0.93], np.deg2rad(1.0), [0.1, -0.1, 0.0])
    H, b, e, n = pb.linearize(st.factor, T)  # cold pass: full search
    # the second and third call at the same pose are warm passes (every certificate holds: no tree walk at all)
    H2, b2, e2, n2 = pb.linearize(st.factor, T)
    H3, b3, e3, n3 = pb.linearize(st.factor, T)
    ps = pb.pass_stats()
    assert ps["cold_passes"] == 1 and ps["warm_passes"] == 2 and ps["walked_points"] < 2000, ps  # only near-ties (runner-up within 1e-5) walk again
    # two warm passes at the same pose: the same pairs; the few points that walked in the first (near-ties) are certified in the second,
    # so their terms move from the walk phase's fp32 wave sums (one point per lane) to the streaming phase's (four): equal to fp32 rounding of
    # a few hundred terms among a million, not bit for bit
    assert np.abs(H2 - H3).max() <= 1e-8 * np.abs(H2).max() and np.abs(b2 - b3).max() <= 1e-8 * np.abs(H2).max() and abs(e2 - e3) <= 1e-8 * e2 and n2 == n3
    # determinism: the same sequence of calls on a fresh problem gives bit-identical sums (no floating-point atomics anywhere)
    pb_again = sga.Problem(c3["tree"], c3["src"])
    Ha, ba, ea, na = pb_again.linearize(st.factor, T)
    Ha2, ba2, ea2, na2 = pb_again.linearize(st.factor, T)
    assert (Ha == H).all() and (ba == b).all() and ea == e and na == n
    assert (Ha2 == H2).all() and (ba2 == b2).all() and ea2 == e2 and na2 == n2
    # cold vs warm: the same neighbours; the two passes run different kernels (one point per lane and row per tile / four points per
    # lane and row per chunk), so the fp32 partial sums group differently
    assert np.abs(H - H2).max() <= 1e-6 * np.abs(H).max() and np.abs(b - b2).max() <= 1e-6 * np.abs(H).max() and abs(e - e2) <= 1e-7 * e and n == n2
    # idempotence of the cached state: the error pass at the linearization point reproduces e
    assert abs(pb.error(st.factor, T) - e) <= 1e-6 * e
    # additivity over source shards (what the multi-GPU all-reduce relies on): halves sum to the whole
    half = len(c3["source"]) // 2
    parts = []
    for sl in (slice(0, half), slice(half, None)):
        sh = sga.PointCloud(c3["source"][sl], covs=c3["src_cov"][sl])
        parts.append(sga.Problem(c3["tree"], sh).linearize(st.factor, T))
    Hs, bs, es, ns = [sum(p[i] for p in parts) for i in range(4)]
    assert ns == n and abs(es - e) <= 1e-9 * e and np.abs(Hs - H).max() <= 1e-7 * np.abs(H).max()  # fp32 wave sums group differently
    # fp32 vs fp64 per-pair arithmetic agree
    st64 = sga.make_setting("GICP", math_mode="fp64")
    H64, b64, e64, n64 = pb.linearize(st64.factor, T)
    assert abs(int(n64) - int(n)) <= 50 and abs(e64 - e) <= 1e-5 * e64 and np.abs(H64 - H).max() <= 1e-5 * np.abs(H64).max()
    # the full registration recovers the synthetic ground truth
    res = pb.align(st)
    dt, dr = pose_error(res.T_target_source, c3["T_gt"])
    assert res.converged and dt < 5e-3 and dr < 5e-4, (dt, dr)


def test_c3_frame_invariance(c3):
    """Moving the target by a rigid G and the pose by G leaves the objective unchanged (up to fp32 rounding of the moved cloud)."""
    st = sga.make_setting("ICP")
    T = np.eye(4)
    e0 = sga.Problem(c3["tree"], c3["src"]).linearize(st.factor, T)[2]
    G = se3([0, 0, 1], np.deg2rad(30), [3.0, -2.0, 0.5])
    tp = c3["tgt"].xyz().astype(np.float64)
    moved = (tp @ G[:3, :3].T + G[:3, 3]).astype(np.float32)
    tree2 = sga.KdTree(sga.PointCloud(moved))
    e1 = sga.Problem(tree2, c3["src"], G).linearize(st.factor, G @ T)[2]
    assert abs(e1 - e0) <= 2e-4 * e0, (e0, e1)


def test_bench_contract():
    """bench.py prints one JSON line with the contract's keys plus roofline and cpu_baseline (small sizes here)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--points", "100000", "--cpu-iters", "3"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["value"] > 0 and j["steps"] >= 20 and j["roofline"]["achieved"] > 0 and j["cpu_baseline"]["value"] > 0
    assert j["final_pose_error"]["trans_m"] < 0.02


def test_scan_to_scan_odometry_matches_oracle(orc):
    """Config C5 protocol on four KITTI-shaped scans (raw scan -> 0.25 m voxel grid -> covariances k = 20 -> GICP against the previous
    scan).  Two comparisons per frame pair:
      (a) the REGISTRATION on identical inputs — the oracle is given the GPU's own downsampled points and covariances — must agree to
          the north-star tolerance, 1e-4 m / 1e-4 rad, with the same iteration count;
      (b) the WHOLE pipeline against the oracle's own preprocessing of the raw scan: 2e-4 m, because the two k = 20 neighbourhoods
          differ for the <= 0.5 % of points whose 20th neighbour is tied within fp32 resolution (test_normals_covariances_match_oracle),
          which moves the optimum by up to ~1e-4 m; and both recover the simulated motion."""
    from small_gicp_amd import odometry

    odom = odometry.OnlineOdometry()
    prev = prev_same = None
    for f in range(4):
        pts, Tws = sga.synthetic.kitti_like_scan(f)
        before = odom.T_world.copy()
        odom.estimate(pts)
        gcloud = odom.target[0]  # the GPU's preprocessed scan
        same = orc.Cloud(gcloud.xyz().astype(np.float64), None, gcloud.covs()[:, :3, :3], tree=True)
        down = orc.voxelgrid_sampling(pts, 0.25)
        cloud = orc.Cloud(down.astype(np.float32).astype(np.float64), tree=True)
        cloud.estimate_normals_covariances(20, 8)
        if prev is not None:
            rel = np.linalg.inv(before) @ odom.T_world
            ref_same = orc.align(prev_same, same, orc.default_setting(factor_kind=orc.GICP, num_threads=8))
            dt, dr = pose_error(rel, ref_same.T_target_source)
            assert dt < POSE_TOL_T and dr < POSE_TOL_R and odom.iterations[-1] == ref_same.iterations + 1, (f, dt, dr, odom.iterations[-1], ref_same.iterations)
            ref = orc.align(prev, cloud, orc.default_setting(factor_kind=orc.GICP, num_threads=8))
            dt, dr = pose_error(rel, ref.T_target_source)
            assert dt < 2e-4 and dr < 1e-4, (f, dt, dr)
            # simulated motion: 1 m forward, 1 deg yaw per frame
            assert abs(np.linalg.norm(rel[:3, 3]) - 1.0) < 0.02
        prev, prev_same = cloud, same


def test_scan_to_scan_odometry_matches_reference():
    """The same C5 protocol against the reference itself (oracle/_ref: the unmodified reference sources compiled in place, protocol of
    src/benchmark/odometry_benchmark_small_gicp_omp.cpp:16-49: voxelgrid_sampling 0.25 m -> KdTree -> estimate_covariances k = 20 ->
    Registration<GICPFactor, ParallelReductionOMP>::align against the previous scan).  (a) registration on identical inputs (the
    reference gets the GPU's preprocessed points and covariances): 1e-4 m / 1e-4 rad and the same iteration count; (b) the whole
    pipeline against the reference's own preprocessing of the raw scan: 2e-4 m (tied 20th neighbours, see the oracle variant above)."""
    from oracle import ref
    from small_gicp_amd import odometry

    if not ref.available():
        pytest.skip("oracle/_ref did not travel with the repository (make -C oracle/ref where /root/reference is mounted)")
    odom = odometry.OnlineOdometry()
    prev = prev_same = None
    worst = [0.0, 0.0, 0.0, 0.0]
    for f in range(5):
        pts, _ = sga.synthetic.kitti_like_scan(f)
        before = odom.T_world.copy()
        odom.estimate(pts)
        gcloud = odom.target[0]
        same = ref.Cloud(gcloud.xyz().astype(np.float64), None, gcloud.covs()[:, :3, :3], tree=True, tree_threads=8)
        down = ref.Cloud(pts.astype(np.float64), tree=False).voxelgrid_sampling(0.25).get()[0]
        cloud = ref.Cloud(down.astype(np.float32).astype(np.float64), tree=True, tree_threads=8)
        cloud.estimate_covariances(20, 8)
        assert abs(len(cloud) - gcloud.size()) == 0  # the voxel grid keeps the same cells
        if prev is not None:
            rel = np.linalg.inv(before) @ odom.T_world
            r_same = ref.align(prev_same, same, ref.GICP, 1.0, 1.0, 8)
            dt, dr = pose_error(rel, r_same.T_target_source)
            worst[0], worst[1] = max(worst[0], dt), max(worst[1], dr)
            assert dt < POSE_TOL_T and dr < POSE_TOL_R and odom.iterations[-1] == r_same.iterations + 1, (f, dt, dr, odom.iterations[-1], r_same.iterations)
            r = ref.align(prev, cloud, ref.GICP, 1.0, 1.0, 8)
            dt, dr = pose_error(rel, r.T_target_source)
            worst[2], worst[3] = max(worst[2], dt), max(worst[3], dr)
            assert dt < 2e-4 and dr < 1e-4, (f, dt, dr)
        prev, prev_same = cloud, same
    print("C5 vs the compiled reference: same inputs %.2e m / %.2e rad, whole pipeline %.2e m / %.2e rad" % tuple(worst))


def test_cpp_odometry_driver_matches_the_python_driver(tmp_path):
    """examples/odometry_benchmark.cpp — the reference's benchmark protocol (odometry_benchmark.cpp + odometry_benchmark_small_gicp_omp.cpp:
    16-49) over include/small_gicp_amd.hpp, with the source entering the registration by its own KdTree — compiled with g++ and run on
    eight KITTI-shaped scans written as .bin files: the trajectory file equals the Python driver's poses (same library calls in the same
    order; the file has six decimals) and the iteration counts agree."""
    from small_gicp_amd import odometry

    cr = odometry.run_synthetic_cpp(8, workdir=str(tmp_path))
    pr = odometry.run_synthetic(8)
    assert len(cr["estimated"]) == 8
    worst = max(np.abs(a - b).max() for a, b in zip(cr["estimated"], pr["estimated"]))
    assert worst < 2e-6, worst
    assert abs(cr["mean_iterations"] - pr["mean_iterations"]) < 0.01, (cr["mean_iterations"], pr["mean_iterations"])  # printed with two decimals
    assert 0 < cr["registration_ms_per_scan"] < 50
    print("C++ odometry driver: %.3f ms/scan registration, %.3f ms/scan total (Python driver: %.3f / %.3f)" % (cr["registration_ms_per_scan"], cr["total_ms_per_scan"], pr["registration_ms_per_scan"], pr["total_ms_per_scan"]))


def test_cpp_header_layer(tmp_path, c1_raw, c1_gold):
    """include/small_gicp_amd.hpp — Registration<Factor, ParallelReductionHIP> and the helper align() overloads, compiled with g++
    against the C-ABI library and run on config C1: same poses as the oracle goldens (1e-4 m / 1e-4 rad)."""
    exe = tmp_path / "test_cpp_api"
    libdir = os.path.dirname(sga.LIB_PATH)
    cmd = ["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_cpp_api.cpp"), "-o", str(exe), "-L" + libdir, "-lsmall_gicp_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    tgt, src, T_gt = c1_raw
    (tmp_path / "t.f32").write_bytes(np.ascontiguousarray(tgt, dtype=np.float32).tobytes())
    (tmp_path / "s.f32").write_bytes(np.ascontiguousarray(src, dtype=np.float32).tobytes())
    p = subprocess.run([str(exe), str(tmp_path / "t.f32"), str(tmp_path / "s.f32")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    cases = {}
    for ln in p.stdout.splitlines():
        tok = ln.split()
        if tok[0] == "CASE":
            cases[tok[1]] = dict(iterations=int(tok[2]), inliers=int(tok[3]), converged=int(tok[4]), error=float(tok[5]), T=np.array([float(x) for x in tok[6:22]]).reshape(4, 4).T)
        elif tok[0] == "SIZES":
            assert [int(tok[1]), int(tok[2])] == c1_gold["downsampled_sizes"]
        elif tok[0] == "VOXELS":
            assert int(tok[1]) == c1_gold["cases"]["VGICP"]["num_voxels"]
        elif tok[0] == "ACCESS":
            assert int(tok[1]) == 1 and int(tok[2]) == 0 and float(tok[3]) < 1e-9 and float(tok[4]) == 1.0 and abs(float(tok[5]) - 2.001) < 1e-3
    for name, gold in [("HELPER_GICP", "GICP"), ("GICP", "GICP"), ("PLANE_ICP", "PLANE_ICP"), ("ICP", "ICP"), ("HUBER_GICP", "HUBER_GICP"), ("CAUCHY_GICP", "CAUCHY_GICP"), ("VGICP", "VGICP"), ("HELPER_VGICP", "VGICP")]:
        g = c1_gold["cases"][gold]
        c = cases[name]
        dt, dr = pose_error(c["T"], np.array(g["T"]))
        assert dt < POSE_TOL_T and dr < POSE_TOL_R, (name, dt, dr)
        assert c["iterations"] == g["iterations"] and abs(c["inliers"] - g["num_inliers"]) <= 2 and c["converged"] == int(g["converged"])
    # Gauss-Newton and the DoF-restricted run land near the ground truth as well
    for name in ("GN_GICP", "RESTRICT_GICP"):
        dt, dr = pose_error(cases[name]["T"], T_gt)
        assert dt < 0.1 and dr < 0.03, (name, dt, dr)
    # RestrictDoF: roll/pitch and z stay (softly) frozen
    # scan-to-model target through the C++ layer == the same calls through ctypes
    fm = sga.IncrementalVoxelMapCov(1.0)
    fm.set_search_offsets(7)
    tgt_c, _ = sga.preprocess_points(c1_raw[0], 0.25, 10)
    src_c, _ = sga.preprocess_points(c1_raw[1], 0.25, 10)
    fm.insert(tgt_c)
    rm = sga.Problem(fm, src_c).align(sga.make_setting("GICP"))
    assert np.abs(rm.T_target_source - cases["MODEL_GICP"]["T"]).max() < 1e-9 and rm.iterations == cases["MODEL_GICP"]["iterations"]
    Tr = cases["RESTRICT_GICP"]["T"]
    assert abs(Tr[2, 3]) < 2e-3 and abs(Tr[2, 0]) < 1e-3 and abs(Tr[2, 1]) < 1e-3  # soft constraints (general_factor.hpp:42)


# Observed on MI355X (round 4, printed with -s): C1 0 of 6147 (k = 10 and 20); the C3 target 1 of 1 000 000 against the compiled reference;
# a C5 scan 0 of 11 441; every other point agrees to 3e-8 (the fp32 rounding of the stored result).  The bounds leave room for a handful.
P2_MAX_DIFFERING_C1 = 2
P2_MAX_DIFFERING_PER_MILLION = 10


def _cpu_features(points32, k, threads):
    """estimate_normals_covariances of the compiled reference (oracle/_ref: util/normal_estimation_omp.hpp over KdTreeBuilderOMP) when it
    travelled with the repository, the oracle's restatement otherwise, on the fp32 points the device holds."""
    from oracle import orc as _orc, ref as _ref

    if _ref.available():
        c = _ref.Cloud(points32.astype(np.float64), tree=True, tree_threads=min(32, threads))
        c.estimate_normals_covariances(k, threads)
        kind = "reference"
    else:
        _orc.build()
        c = _orc.Cloud(points32.astype(np.float64))
        c.estimate_normals_covariances(k, threads)
        kind = "port"
    _, on, ocov = c.get()
    return on, ocov, kind


def test_covariances_at_scale_match_reference():
    """p2 at the size of C3 and on a C5 scan: local_features_kernel (k = 20; the LDS-window path of large clouds) against
    the reference's own estimate_normals_covariances_omp on the same fp32 points.  The number of differing neighbourhoods is printed and
    bounded; every other point agrees to 1e-5."""
    threads = max(1, min(64, os.cpu_count() or 1))
    target = sga.synthetic.scene(1_000_000, 1)
    scan, _ = sga.synthetic.kitti_like_scan(3)
    down = sga.voxelgrid_sampling(scan, 0.25).xyz().astype(np.float32)
    for name, pts in (("C3 target (1M)", target), ("C5 scan after the 0.25 m grid", down)):
        cloud = sga.PointCloud(pts)
        sga.estimate_normals_covariances(cloud, None, 20)
        gn, gcov = cloud.normals()[:, :3], cloud.covs()[:, :3, :3]
        on, ocov, kind = _cpu_features(pts, 20, threads)
        bad, worst_n, worst_c = _neighbourhood_differences(gn, gcov, on, ocov)
        per_million = 1e6 * bad.sum() / len(pts)
        print("%s vs %s: %d of %d neighbourhoods differ (%.0f per million); the others agree to %.1e (normals) / %.1e (covariances)" % (name, kind, int(bad.sum()), len(pts), per_million, worst_n, worst_c))
        assert (per_million <= P2_MAX_DIFFERING_PER_MILLION or bad.sum() <= P2_MAX_DIFFERING_C1) and worst_n < 1e-5 and worst_c < 1e-5, (name, int(bad.sum()), worst_n, worst_c)


def test_c3_with_cpu_estimated_covariances():
    """The two stages pinned independently: the covariances of a C3-shaped pair (300k points) estimated by the CPU
    reference, handed to BOTH sides; linearization at two poses and the registration against the CPU."""
    from oracle import orc as _orc, ref as _ref

    _orc.build()
    threads = max(1, min(64, os.cpu_count() or 1))
    target, source, T_gt = sga.synthetic.registration_pair(300_000)
    _, tcov, _ = _cpu_features(target, 20, threads)
    _, scov, _ = _cpu_features(source, 20, threads)
    tgt = sga.PointCloud(target, None, tcov)
    src = sga.PointCloud(source, None, scov)
    pb = sga.Problem(sga.KdTree(tgt), src)
    # the device rounds the covariances to fp32: the CPU gets those very numbers
    tc32, sc32 = tgt.covs()[:, :3, :3], src.covs()[:, :3, :3]
    otc = _orc.Cloud(target.astype(np.float64), None, tc32, tree=True)
    osc = _orc.Cloud(source.astype(np.float64), None, sc32, tree=False)
    st = sga.make_setting("GICP")
    for T in (np.eye(4), T_gt @ se3([0.3, -0.5, 0.8], np.deg2rad(0.05), [0.004, -0.003, 0.002])):
        f = _orc.Factors(len(osc))
        Ho, bo, eo, no = _orc.linearize(otc, osc, _orc.default_setting(factor_kind=_orc.GICP, num_threads=threads, max_dist_sq=1.0), T, f)
        H, b, e, n = pb.linearize(st.factor, T)
        assert abs(int(n) - int(no)) <= 3 and np.abs(H - Ho).max() <= FP32_REL * np.abs(Ho).max() and abs(e - eo) <= FP32_REL * eo
    res = pb.align(st)
    if _ref.available():
        r = _ref.align(_ref.Cloud(target.astype(np.float64), None, tc32, tree=True, tree_threads=min(32, threads)), _ref.Cloud(source.astype(np.float64), None, sc32, tree=False), _ref.GICP, 1.0, 1.0, threads)
    else:
        r = _orc.align(otc, osc, _orc.default_setting(factor_kind=_orc.GICP, num_threads=threads))
    dt, dr = pose_error(res.T_target_source, r.T_target_source)
    print("C3-shaped registration on CPU-estimated covariances: dt %.2e m, dr %.2e rad, iterations %d / %d" % (dt, dr, res.iterations, r.iterations))
    assert dt < POSE_TOL_T and dr < POSE_TOL_R and res.iterations == r.iterations


def test_c2_plane_icp_100k_matches_oracle(orc):
    """Config C2 (point-to-plane ICP, 100k <-> 100k synthetic fp32 points, normals k = 20): accumulators at two poses and the final
    pose against the oracle fed the very same fp32 points and normals."""
    target, source, T_gt = sga.synthetic.registration_pair(100_000)
    tgt = sga.PointCloud(target)
    sga.estimate_normals(tgt, None, 20)
    tree = sga.KdTree(tgt)
    src = sga.PointCloud(source)
    nrm = tgt.normals()[:, :3]
    otc = orc.Cloud(target.astype(np.float64), nrm)
    osc = orc.Cloud(source.astype(np.float64), tree=False)
    st = sga.make_setting("PLANE_ICP")
    os_ = orc.default_setting(factor_kind=orc.PLANE_ICP, num_threads=8)
    pb = sga.Problem(tree, src)
    f = orc.Factors(len(osc))
    for T in (np.eye(4), T_gt):
        H, b, e, n = pb.linearize(st.factor, T)
        Ho, bo, eo, no = orc.linearize(otc, osc, os_, T, f)
        assert abs(int(n) - int(no)) <= 5
        assert np.abs(H - Ho).max() <= FP32_REL * np.abs(Ho).max() and abs(e - eo) <= FP32_REL * eo
    res = pb.align(st)
    ref = orc.align(otc, osc, os_)
    dt, dr = pose_error(res.T_target_source, ref.T_target_source)
    assert dt < POSE_TOL_T and dr < POSE_TOL_R and res.iterations == ref.iterations, (dt, dr, res.iterations, ref.iterations)
    dt, dr = pose_error(res.T_target_source, T_gt)
    assert dt < 0.02 and dr < 2e-3


def test_c3_in_a_geo_referenced_frame_is_c3(c3):
    """Config C3 with both clouds moved by (25 600, -51 200, 128) m — whole multiples of 128 m, so the device records are the unshifted ones
    bit for bit (small_gicp_amd.h: device frames) and the whole 10-pass registration (cold, warm, queue-fed and streaming kernels, their
    motion thresholds, the certificates, the error model) must reproduce itself: the same passes, the same inliers, the pose conjugated by
    the shift.  The C1-sized frame tests are in tests/test_coordinate_range.py; this is the benchmark's size."""
    s = np.array([25600.0, -51200.0, 128.0])
    tgt, src = c3["tgt"], c3["src"]
    tc, sc = tgt.covs()[:, :3, :3], src.covs()[:, :3, :3]
    tgt2 = sga.PointCloud(tgt.xyz().astype(np.float64) + s, None, tc)
    src2 = sga.PointCloud(src.xyz().astype(np.float64) + s, None, sc)
    assert (tgt2.origin() == s).all() and (src2.origin() == s).all()
    st = sga.make_setting("GICP", max_iterations=10, rotation_eps=0.0, translation_eps=0.0)  # the benchmark's protocol: ten iterations
    p1, p2 = sga.Problem(c3["tree"], src), sga.Problem(sga.KdTree(tgt2), src2)
    r1, r2 = p1.align(st), p2.align(st)
    T1s = r1.T_target_source.copy()
    T1s[:3, 3] = r1.T_target_source[:3, 3] + s - r1.T_target_source[:3, :3] @ s
    c = np.append(src.xyz().astype(np.float64).mean(axis=0) + s, 1.0)
    d = float(np.linalg.norm((T1s @ c - r2.T_target_source @ c)[:3]))
    print("C3 in a frame 57 km from the origin: %.2e m at the data from the unshifted registration, iterations %d / %d, inliers %d / %d, passes %s" % (d, r2.iterations, r1.iterations, r2.num_inliers, r1.num_inliers, p2.pass_stats()))
    assert r1.iterations == r2.iterations and r1.num_inliers == r2.num_inliers and d < 1e-6
    s1, s2 = p1.pass_stats(), p2.pass_stats()
    assert s1["cold_passes"] == s2["cold_passes"] and s1["warm_passes"] == s2["warm_passes"]
    differ = int((p1.factors()[0] != p2.factors()[0]).sum())
    print("correspondences differing between the two frames: %d of 1M (the pose reaches the kernels through one more rounding)" % differ)
    assert differ <= 10  # observed: 1


# ---- configs C3 and C4 at their full size against the reference itself ------------------------------------------------------------
@pytest.fixture(scope="module")
def c3_cpu(c3):
    """The C3 clouds as the CPU sees them: the GPU's fp32 points and covariances handed to the compiled reference (oracle/_ref, when it
    travelled with the repository) and to the oracle's restatement (for per-point factor state, which the reference's C API does not
    export; the two agree to 1e-9, tests/test_oracle_vs_reference.py)."""
    from oracle import orc as _orc, ref as _ref

    _orc.build()
    tp, sp = c3["tgt"].xyz().astype(np.float64), c3["src"].xyz().astype(np.float64)
    tcov, scov = c3["tgt"].covs()[:, :3, :3], c3["src"].covs()[:, :3, :3]
    threads = max(1, min(64, os.cpu_count() or 1))
    out = dict(orc=_orc, threads=threads, otc=_orc.Cloud(tp, None, tcov, tree=True), osc=_orc.Cloud(sp, None, scov, tree=False), ref=None, src_xyz64=sp)
    if _ref.available():
        out.update(ref=_ref, rtc=_ref.Cloud(tp, None, tcov, tree=True, tree_threads=min(32, threads)), rsc=_ref.Cloud(sp, None, scov, tree=False))
    return out


# Observed on MI355X (round 3, printed by the test with -s): at the identity 0 of 1M correspondences differ from the reference's, at the
# near-converged pose 12 (queries with two candidates within fp32 rounding of each other: the device compares fp32 distances, the
# reference doubles); inlier deltas 0 in every case.  The asserts allow about twice the observed numbers.
C3_MAX_DIFFERING_PAIRS = 30
C3_MAX_INLIER_DELTA = 4


def test_c3_matches_reference(c3, c3_cpu):
    """Config C3 (GICP, 1M <-> 1M) at size: one linearization at two poses (H / b / e to 2e-5, at most 30 of the million correspondences differ)
    and the whole registration (pose 1e-4 m / 1e-4 rad, equal iteration count) against the reference's own code —
    registration_helper.cpp:81-137 align(), Registration<GICPFactor, ParallelReductionOMP> — on the same inputs.  At this size a few
    hundred of the million queries have two candidates tied within fp32 resolution; the device compares fp32 distances in both math
    modes, the reference doubles, so those pairs differ and the fp64-math sums agree to ~3e-6 rather than the 1e-10 of config C1."""
    orc, threads = c3_cpu["orc"], c3_cpu["threads"]
    pb = sga.Problem(c3["tree"], c3["src"])
    poses = [np.eye(4), c3["T_gt"] @ se3([0.3, -0.5, 0.8], np.deg2rad(0.05), [0.004, -0.003, 0.002])]
    for T in poses:
        f = orc.Factors(len(c3_cpu["osc"]))
        Ho, bo, eo, no = orc.linearize(c3_cpu["otc"], c3_cpu["osc"], orc.default_setting(factor_kind=orc.GICP, num_threads=threads, max_dist_sq=1.0), T, f)
        if c3_cpu["ref"] is not None:  # the compiled reference has the last word on the sums
            Hr, br, er, _, nr = c3_cpu["ref"].linearize(c3_cpu["rtc"], c3_cpu["rsc"], c3_cpu["ref"].GICP, 0, 1.0, 1.0, threads, T)
            assert np.abs(Hr - Ho).max() <= 1e-9 * np.abs(Ho).max() and nr == no
            Ho, bo, eo = Hr, br, er
        for mode, rel in (("fp32", FP32_REL), ("fp64", FP32_REL)):
            H, b, e, n = pb.linearize(sga.make_setting("GICP", math_mode=mode).factor, T)
            assert np.abs(H - Ho).max() <= rel * np.abs(Ho).max(), (mode, np.abs(H - Ho).max() / np.abs(Ho).max())
            assert np.abs(b - bo).max() <= rel * max(np.abs(bo).max(), 1e-3 * np.abs(Ho).max()) and abs(e - eo) <= rel * eo
            print("C3 inliers %s: %d (reference %d, delta %+d)" % (mode, n, no, int(n) - int(no)))
            assert abs(int(n) - int(no)) <= (C3_MAX_INLIER_DELTA if mode == "fp32" else 2), (mode, n, no)  # pairs within fp32 rounding of the 1 m rejector
        got = pb.factors()[0]
        want = f.get()[0]
        if c3_cpu["ref"] is not None:  # the reference's own nearest_neighbor_search + DistanceRejector (ann/kdtree.hpp:193-205, rejector.hpp:19-28)
            q = c3_cpu["src_xyz64"] @ T[:3, :3].T + T[:3, 3]
            idx, d2 = c3_cpu["rtc"].nearest(q, threads)
            want_ref = np.where(d2 > 1.0, -1, idx)
            assert (want_ref != want).sum() <= 2, "the restatement's correspondences differ from the reference's: %d" % (want_ref != want).sum()
            want = want_ref
        # the reference numbers target points in the caller's order, so do we
        differing = int((got != want).sum())
        print("C3 correspondences differing from the reference: %d of %d" % (differing, len(want)))
        assert differing <= C3_MAX_DIFFERING_PAIRS, differing
    for mode in ("fp32", "fp64"):
        res = pb.align(sga.make_setting("GICP", math_mode=mode))
        if c3_cpu["ref"] is not None:
            r = c3_cpu["ref"].align(c3_cpu["rtc"], c3_cpu["rsc"], c3_cpu["ref"].GICP, 1.0, 1.0, threads)
        else:
            r = orc.align(c3_cpu["otc"], c3_cpu["osc"], orc.default_setting(factor_kind=orc.GICP, num_threads=threads))
        dt, dr = pose_error(res.T_target_source, r.T_target_source)
        assert dt < POSE_TOL_T and dr < POSE_TOL_R and res.iterations == r.iterations and res.converged == r.converged, (mode, dt, dr, res.iterations, r.iterations)
        print("C3 registration %s: dt %.2e m, dr %.2e rad, iterations %d / %d, inliers %d / %d" % (mode, dt, dr, res.iterations, r.iterations, res.num_inliers, r.num_inliers))
        assert abs(int(res.num_inliers) - int(r.num_inliers)) <= C3_MAX_INLIER_DELTA


def test_c4_matches_reference(c3, c3_cpu):
    """Config C4 (VGICP: GaussianVoxelMap(0.5 m) of the 1M target, 1M source points) at size: the voxel map itself, one linearization
    and the whole registration against the CPU (the reference's align(GaussianVoxelMap, ...) of registration_helper.cpp:125-137 when
    oracle/_ref is present, the oracle's restatement otherwise)."""
    orc, threads = c3_cpu["orc"], c3_cpu["threads"]
    vm = sga.GaussianVoxelMap(0.5)
    vm.insert(c3["tgt"])
    ovm = orc.VoxelMap(c3_cpu["otc"], 0.5)
    coords, means, c6, counts = vm.download()
    oc, om, ocv, ocnt = ovm.get()
    assert len(coords) == len(oc) and (coords == oc).all() and (counts == ocnt).all()  # same voxels in the same (first-touch) order
    assert np.abs(means - om).max() < 1e-5 * max(1.0, np.abs(om).max())
    pb = sga.Problem(vm, c3["src"])
    T = c3["T_gt"] @ se3([0.3, -0.5, 0.8], np.deg2rad(0.05), [0.004, -0.003, 0.002])
    f = orc.Factors(len(c3_cpu["osc"]))
    Ho, bo, eo, no = orc.linearize(ovm, c3_cpu["osc"], orc.default_setting(factor_kind=orc.GICP, num_threads=threads, max_dist_sq=1.0), T, f)
    for mode in ("fp32", "fp64"):  # the voxel means and covariances themselves are rounded to fp32 on the device: both modes to 2e-5
        H, b, e, n = pb.linearize(sga.make_setting("GICP", math_mode=mode).factor, T)
        assert np.abs(H - Ho).max() <= FP32_REL * np.abs(Ho).max(), (mode, np.abs(H - Ho).max() / np.abs(Ho).max())
        ti, _ = pb.factors()
        oti, _ = f.get(is_voxelmap=True)
        differ = int((ti != oti).sum())
        print("C4 %s: %d of %d voxel assignments differ from the CPU's (queries within fp32 rounding of a voxel face), inliers %d / %d" % (mode, differ, len(ti), n, no))
        assert abs(e - eo) <= FP32_REL * eo and abs(int(n) - int(no)) <= 4 and differ <= 20, (n, no, differ)  # observed: 7 / 5 assignments, 1 / 0 inliers
    res = pb.align(sga.make_setting("GICP"))
    if c3_cpu["ref"] is not None:
        r = c3_cpu["ref"].align(c3_cpu["rtc"], c3_cpu["rsc"], c3_cpu["ref"].VGICP, 0.5, 1.0, threads)
    else:
        r = orc.align(ovm, c3_cpu["osc"], orc.default_setting(factor_kind=orc.GICP, num_threads=threads))
    dt, dr = pose_error(res.T_target_source, r.T_target_source)
    assert dt < POSE_TOL_T and dr < POSE_TOL_R and res.iterations == r.iterations, (dt, dr, res.iterations, r.iterations)


def test_c4_vgicp_1m_properties(c3):
    """Config C4 (VGICP: GaussianVoxelMap(0.5 m) of the 1M target, 1M source points): determinism, idempotence of the cached state,
    shard additivity, agreement of fp32 and fp64 math, and recovery of the ground truth."""
    vm = sga.GaussianVoxelMap(0.5)
    vm.insert(c3["tgt"])
    coords, means, c6, counts = vm.download()
    assert counts.sum() == 1_000_000 and len(np.unique(coords, axis=0)) == len(coords)
    # every voxel mean lies inside its voxel
    assert (np.floor(means.astype(np.float64) / 0.5).astype(np.int64) == coords).mean() > 0.9999
    st = sga.make_setting("GICP")
    pb = sga.Problem(vm, c3["src"])
    T = c3["T_gt"]
    H, b, e, n = pb.linearize(st.factor, T)
    H2, b2, e2, n2 = pb.linearize(st.factor, T)
    assert (H == H2).all() and e == e2 and n == n2 and n > 500_000
    assert abs(pb.error(st.factor, T) - e) <= 1e-6 * e
    half = len(c3["source"]) // 2
    parts = [sga.Problem(vm, sga.PointCloud(c3["source"][sl], covs=c3["src_cov"][sl])).linearize(st.factor, T) for sl in (slice(0, half), slice(half, None))]
    assert sum(p[3] for p in parts) == n and abs(sum(p[2] for p in parts) - e) <= 1e-9 * e
    H64, b64, e64, n64 = pb.linearize(sga.make_setting("GICP", math_mode="fp64").factor, T)
    assert abs(int(n64) - int(n)) <= 50 and abs(e64 - e) <= 1e-5 * e64
    res = pb.align(st)
    dt, dr = pose_error(res.T_target_source, c3["T_gt"])
    assert res.converged and dt < 0.02 and dr < 2e-3, (dt, dr)


# ---- the search structure itself ------------------------------------------------------------------------------------------------
def test_kd_build_paths_give_the_same_tree(monkeypatch):
    """Bottom levels finished in LDS (kd_finish_kernel) vs every level through the global radix-sort path: identical trees, hence
    bit-identical linearizations and identical kNN answers (index_build.hip).  Small clouds are built by a third path — one launch per
    level, radix select + partition (kd_split_level_kernel): another valid tree over the same points (the order inside a half is not
    the sorted one), so the same neighbour DISTANCES, inliers and sums to rounding."""
    rng = np.random.default_rng(11)
    for n in (1500, 20_000, 40_000, 600_000):  # one workgroup / the split path's range / small-cloud capacity / large-cloud capacity
        target, source, _ = sga.synthetic.registration_pair(n)
        target = target.copy()
        target[: n // 50] = target[n // 50 : 2 * (n // 50)]  # duplicates: the stable tie order must be reproduced too
        st = sga.make_setting("ICP")
        T = se3([0.2, 0.3, 0.93], np.deg2rad(1.0), [0.1, -0.1, 0.0])
        q = np.concatenate([source[rng.choice(n, 300, replace=False)], rng.uniform(-60, 60, (100, 3)).astype(np.float32)])

        def run():
            tree = sga.KdTree(sga.PointCloud(target))
            H, b, e, inl = sga.Problem(tree, sga.PointCloud(source)).linearize(st.factor, T)
            idx, d2 = tree.batch_knn_search(q, 10)
            return H, b, e, inl, idx, d2

        out = []
        monkeypatch.setenv("SGA_KD_SPLIT", "0")
        for finish in ("1", "0"):
            monkeypatch.setenv("SGA_KD_FINISH", finish)
            out.append(run())
        a, c = out
        assert (a[0] == c[0]).all() and (a[1] == c[1]).all() and a[2] == c[2] and a[3] == c[3], n
        assert (a[4] == c[4]).all() and (a[5] == c[5]).all(), n
        monkeypatch.delenv("SGA_KD_SPLIT")
        monkeypatch.setenv("SGA_KD_FINISH", "1")
        s = run()  # n <= 32768: the split path
        assert s[3] == a[3] and (s[5] == a[5]).all(), n
        assert np.abs(s[0] - a[0]).max() <= 1e-5 * np.abs(a[0]).max() and abs(s[2] - a[2]) <= 1e-5 * abs(a[2]), n


def test_kd_split_path_on_degenerate_clouds(monkeypatch):
    """kd_split_level_kernel (radix select + partition, clouds <= 32768 points) on inputs that stress the select: all points identical, a
    line, two distinct points, a lattice with triplicates, coordinates up to 1e6 with ties, and sizes around the kernel's segment classes
    (256 / 1024 threads x 2 ... 32 keys).  The kNN DISTANCES must equal those of the sort-based build (another valid tree over the same
    points) and brute force."""
    rng = np.random.default_rng(21)
    g = np.arange(16, dtype=np.float32)
    lattice = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    clouds = {
        "identical": np.tile(np.array([[1.5, -2.0, 0.25]], np.float32), (5000, 1)),
        "line": np.c_[rng.uniform(-100, 100, 9000), np.zeros(9000), np.zeros(9000)].astype(np.float32),
        "two points": np.array([[0, 0, 0], [1, 1, 1]], np.float32)[rng.integers(0, 2, 3000)],
        "lattice x3": np.concatenate([lattice, lattice, lattice]),
        "large range with ties": (rng.integers(-1000, 1000, (20000, 3)) * 1000.0).astype(np.float32),
    }
    for n in (255, 257, 1023, 1025, 4097, 8193, 16385, 32767, 32768):
        clouds["uniform %d" % n] = rng.uniform(-10, 10, (n, 3)).astype(np.float32)
    for name, pts in clouds.items():
        q = np.concatenate([pts[rng.choice(len(pts), min(200, len(pts)), replace=False)] + rng.normal(0, 0.3, (min(200, len(pts)), 3)).astype(np.float32), rng.uniform(-20, 20, (50, 3)).astype(np.float32)])
        k = min(5, len(pts))
        res = []
        for split in ("1", "0"):
            monkeypatch.setenv("SGA_KD_SPLIT", split)
            tree = sga.KdTree(sga.PointCloud(pts))
            res.append(tree.batch_knn_search(q, k)[1])
        monkeypatch.delenv("SGA_KD_SPLIT")
        assert np.array_equal(res[0], res[1]), name
        d = ((q[:, None, :].astype(np.float64) - pts[None, : min(len(pts), 40000), :].astype(np.float64)) ** 2).sum(-1)
        brute = np.sort(d, axis=1)[:, :k]
        assert np.abs(res[0] - brute).max() <= 1e-5 * max(1.0, float(brute.max())), name


def test_kd_top_levels_on_degenerate_clouds(monkeypatch):
    """The levels above the split kernel's reach (segments of more than 32768 points: kd_top_*_kernel — select + partition with a segment
    spread over many workgroups, the points moving with the permutation) on inputs that stress the select across chunks: all points
    identical, a line, two distinct points, a lattice with many copies (median keys shared by thousands of points in several chunks),
    large coordinates with ties, and sizes around the chunk (4096) and segment boundaries.  The kNN DISTANCES must equal those of the
    sort-based levels (SGA_KD_TOP=0: another valid tree over the same points) and brute force."""
    rng = np.random.default_rng(33)
    g = np.arange(16, dtype=np.float32)
    lattice = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    clouds = {
        "identical": np.tile(np.array([[1.5, -2.0, 0.25]], np.float32), (70001, 1)),
        "line": np.c_[rng.uniform(-100, 100, 90000), np.zeros(90000), np.zeros(90000)].astype(np.float32),
        "two points": np.array([[0, 0, 0], [1, 1, 1]], np.float32)[rng.integers(0, 2, 66000)],
        "lattice x20": np.concatenate([lattice] * 20),
        "large range with ties": (rng.integers(-1000, 1000, (150000, 3)) * 1000.0).astype(np.float32),
    }
    for n in (32769, 40961, 65535, 65537, 131073, 270001):
        clouds["uniform %d" % n] = rng.uniform(-10, 10, (n, 3)).astype(np.float32)
    for name, pts in clouds.items():
        q = np.concatenate([pts[rng.choice(len(pts), 150, replace=False)] + rng.normal(0, 0.3, (150, 3)).astype(np.float32), rng.uniform(-20, 20, (50, 3)).astype(np.float32)])
        k = 5
        res = []
        for top in ("1", "0"):
            monkeypatch.setenv("SGA_KD_TOP", top)
            tree = sga.KdTree(sga.PointCloud(pts))
            res.append(tree.batch_knn_search(q, k)[1])
        monkeypatch.delenv("SGA_KD_TOP")
        assert np.array_equal(res[0], res[1]), name
        brute = np.empty((len(q), k))
        p64 = pts.astype(np.float64)
        for a in range(0, len(q), 20):  # (chunks of queries: 270k x 20 x 3 doubles at a time)
            d = ((q[a : a + 20, None, :].astype(np.float64) - p64[None, :, :]) ** 2).sum(-1)
            brute[a : a + 20] = np.sort(np.partition(d, k, axis=1)[:, :k], axis=1)
        assert np.abs(res[0] - brute).max() <= 1e-5 * max(1.0, float(brute.max())), name


def test_nearest_neighbour_exact_at_scale_and_seed_independent(c3):
    """The registration search (pair records + plane and box pruning + seeds from the previous pose) returns the exact nearest
    neighbour at the C3 size: checked against brute force on a sample, and seeded == unseeded on all 1M correspondences."""
    st = sga.make_setting("ICP", max_correspondence_distance=1.0)
    T0 = np.eye(4)
    T1 = se3([0.2, 0.3, 0.93], np.deg2rad(1.5), [0.2, -0.15, 0.03])
    pb = sga.Problem(c3["tree"], c3["src"])
    pb.linearize(st.factor, T0)          # fills the seeds with the neighbours at T0
    seeded = pb.linearize(st.factor, T1)
    corr_seeded = pb.factors()[0]
    pb2 = sga.Problem(c3["tree"], c3["src"])
    fresh = pb2.linearize(st.factor, T1)  # no seeds
    corr_fresh = pb2.factors()[0]
    differ = corr_seeded != corr_fresh
    assert differ.mean() < 1e-5  # only exact distance ties may resolve differently
    assert abs(seeded[2] - fresh[2]) <= 1e-9 * fresh[2] and seeded[3] == fresh[3]
    # brute force on a sample (float64 distances; fp32 near-ties tolerated)
    tp = c3["tgt"].xyz().astype(np.float64)
    sp = c3["source"].astype(np.float64)
    rng = np.random.default_rng(5)
    sample = rng.choice(len(sp), 600, replace=False)
    qs = sp[sample] @ T1[:3, :3].T + T1[:3, 3]
    ok = 0
    for k, i in enumerate(sample):
        d2 = ((tp - qs[k]) ** 2).sum(1)
        j = int(np.argmin(d2))
        got = int(corr_fresh[i])
        if d2[j] > 1.0:
            ok += got == -1 or abs(d2[got] - 1.0) < 1e-5
        else:
            ok += got == j or (got >= 0 and d2[got] - d2[j] <= 1e-6 * max(d2[j], 1e-3))
    assert ok == len(sample), ok


# ---- §8f rows 1-2: the reference's Python module name / signatures and the on-disk formats ------------------------------------------
def test_small_gicp_module_surface(tmp_path, c1_raw, c1_gold):
    """`import small_gicp` (the reference's module name) drives the GPU engine with the binding's signatures: the checks of the
    reference's src/test/python_test.py, restated (load -> preprocess -> the align overloads -> per-point factors)."""
    import small_gicp
    from small_gicp_amd import io

    raw_t, raw_s, T_gt = c1_raw
    io.write_ply(tmp_path / "target.ply", raw_t)
    io.write_ply(tmp_path / "source.ply", raw_s)
    target_raw = small_gicp.read_ply(str(tmp_path / "target.ply"))
    source_raw = small_gicp.read_ply(str(tmp_path / "source.ply"))
    tnp, snp = target_raw.points(), source_raw.points()
    assert tnp.shape == (len(raw_t), 4) and snp.shape[1] == 4 and (tnp[:, 3] == 1).all()

    def check(result, tol_t=0.05, tol_r=0.05):  # python_test.py:52-58
        dt, dr = pose_error(result.T_target_source, T_gt)
        assert dt < tol_t and dr < tol_r, (dt, dr)

    # preprocess: numpy (Nx4 and Nx3) and PointCloud inputs, downsampled sizes of config C1
    target, target_tree = small_gicp.preprocess_points(tnp, downsampling_resolution=0.25)
    source, source_tree = small_gicp.preprocess_points(snp[:, :3], downsampling_resolution=0.25)
    assert [target.size(), source.size()] == c1_gold["downsampled_sizes"]
    t2, _ = small_gicp.preprocess_points(target_raw, downsampling_resolution=0.25)
    assert t2.size() == target.size() and np.abs(t2.points() - target.points()).max() < 1e-6
    assert np.abs(np.linalg.norm(target.normals()[:, :3], axis=1) - 1).max() < 1e-5 and target.covs().shape == (target.size(), 4, 4)
    # overload 1: raw numpy clouds, keyword arguments
    check(small_gicp.align(tnp, snp, downsampling_resolution=0.25))
    for rtype in ("ICP", "PLANE_ICP", "GICP", "VGICP"):
        check(small_gicp.align(tnp, snp, np.eye(4), rtype, 1.0, 0.25, 1.0, 4, 20), 0.2, np.deg2rad(2.5))
    # overload 2: preprocessed clouds + tree, positional init guess
    result = small_gicp.align(target, source, target_tree)
    check(result)
    result2 = small_gicp.align(target, source, target_tree, result.T_target_source)
    check(result2)
    assert result.converged and result.num_inliers > 0.8 * source.size() and result.H.shape == (6, 6) and result.b.shape == (6,)
    # overload 3: voxel map target
    vm = small_gicp.GaussianVoxelMap(1.0)
    vm.insert(target)
    check(small_gicp.align(vm, source))
    # step by step (basic_registration.py:91-119)
    t3 = small_gicp.voxelgrid_sampling(target_raw, 0.25)
    s3 = small_gicp.voxelgrid_sampling(source_raw, 0.25)
    tree3 = small_gicp.KdTree(t3, num_threads=4)
    small_gicp.estimate_covariances(t3, tree3)
    small_gicp.estimate_covariances(s3, small_gicp.KdTree(s3))
    check(small_gicp.align(t3, s3, tree3))
    # per-point factors: their sum is the linearized system of the registration at that pose (python_test.py:143-166, 5 %)
    factor, rejector = small_gicp.GICPFactor(), small_gicp.DistanceRejector()
    H = np.zeros((6, 6))
    n_ok = 0
    for i in range(source.size()):
        ok, Hi, bi, ei = factor.linearize(target, source, target_tree, result2.T_target_source, i, rejector)
        if ok:
            H += Hi
            n_ok += 1
    assert n_ok == result2.num_inliers or abs(n_ok - result2.num_inliers) < 20
    assert np.max(np.abs(result2.H - H) / np.abs(result2.H).max()) < 0.05
    # ... and equals the GPU linearization at the same pose far tighter than that
    st = sga.make_setting("GICP")
    Hg = sga.Problem(target_tree, source).linearize(st.factor, result2.T_target_source)[0]
    assert np.abs(Hg - H).max() <= 1e-4 * np.abs(Hg).max()


def test_kitti_driver_and_formats(tmp_path, capsys):
    """Directory of KITTI .bin scans -> odometry driver -> trajectory file (odometry_benchmark.cpp:20-97), same poses as the in-memory run."""
    from small_gicp_amd import io, odometry

    frames = 5
    ref = odometry.run_synthetic(frames)
    d = tmp_path / "velodyne"
    d.mkdir()
    for f in range(frames):
        pts, _ = sga.synthetic.kitti_like_scan(f)
        io.write_points(d / ("%06d.bin" % f), pts)
    (d / "ignored.txt").write_text("not a scan")
    odometry.main([str(d), str(tmp_path / "traj.txt")])
    out = capsys.readouterr().out
    assert "num_frames=%d" % frames in out and "registration_time_stats=" in out and "[msec/scan]" in out
    traj = io.read_trajectory(tmp_path / "traj.txt")
    assert len(traj) == frames
    for T, E in zip(traj, ref["estimated"]):
        assert np.abs(T[:3] - E[:3]).max() < 5e-6  # "%.6f"


def test_pipelined_odometry_gives_the_same_poses():
    """Two contexts (streams) + a producer thread: preprocessing of frame i+1 overlaps the registration of frame i; the poses are
    bit-identical to the sequential driver's."""
    from small_gicp_amd import odometry

    seq = odometry.run_synthetic(6)
    pipe = odometry.run_synthetic_pipelined(6)
    assert len(pipe["estimated"]) == 6
    for a, b in zip(seq["estimated"], pipe["estimated"]):
        assert (a == b).all()


@pytest.mark.parametrize("workers,reg_workers,frames", [(2, 2, 9), (3, 3, 10), (1, 4, 7), (2, 2, 1), (2, 3, 2)])
def test_flow_odometry_registers_pairs_side_by_side_and_gives_the_same_poses(workers, reg_workers, frames):
    """The flow form of the reference's TBB engine (odometry_benchmark_small_gicp_tbb_flow.cpp:55-141: preprocessing AND registration nodes with
    unlimited concurrency, sequencers in between): pairs are registered from the identity, so several run at once on contexts of their own;
    the products in frame order are bit-identical to the sequential driver's, whatever order the pairs finish in."""
    from small_gicp_amd import odometry

    seq = odometry.run_synthetic(frames)
    pipe = odometry.run_synthetic_pipelined(frames, workers=workers, reg_workers=reg_workers)
    assert len(pipe["estimated"]) == frames
    for a, b in zip(seq["estimated"], pipe["estimated"]):
        assert (a == b).all()
    if frames > 1:
        assert abs(pipe["mean_iterations"] - seq["mean_iterations"]) < 1e-12


def test_cpp_flow_driver_matches_the_sequential_cpp_driver(tmp_path):
    """examples/odometry_benchmark_flow.cpp — std::thread workers with a context (HIP stream) each over include/small_gicp_amd.hpp
    (PointCloud on a named context, ParallelReductionHIP::context), pinned and pageable scans: the trajectory file equals the sequential
    C++ driver's line for line, the iteration counts agree."""
    from small_gicp_amd import odometry

    cr = odometry.run_synthetic_cpp(10, workdir=str(tmp_path))
    for pw, rw, pinned in ((2, 2, False), (3, 2, True), (1, 1, False)):
        fr = odometry.run_synthetic_cpp_flow(10, workdir=str(tmp_path), preprocess_workers=pw, registration_workers=rw, pinned=pinned, repeat=2)
        assert len(fr["estimated"]) == 10
        worst = max(np.abs(a - b).max() for a, b in zip(cr["estimated"], fr["estimated"]))
        assert worst == 0.0, (pw, rw, pinned, worst)  # the same numbers printed with the same format
        assert abs(fr["mean_iterations"] - cr["mean_iterations"]) < 0.01
        print("C++ flow driver %d x %d%s: %.3f ms/scan (sequential driver: %.3f total)" % (pw, rw, " pinned" if pinned else "", fr["ms_per_scan"], cr["total_ms_per_scan"]))


# ---- §8f row 3: incremental GaussianVoxelMap (scan-to-model target) -----------------------------------------------------------------
def test_incremental_voxelmap_matches_oracle(orc, c1_f32, gpu_c1):
    """The same sequence of insert(cloud, T) on the device (csrc/voxelmap.hip) and in the oracle (pinned to the reference's
    IncrementalVoxelMap in tests/test_oracle_vs_reference.py): voxel creation order, counts, running means / covariances, and the
    LRU sweep.  Inputs are the identical fp32 clouds; the map state is fp64 on both sides, the device exports fp32."""
    d = c1_f32
    tgt, src, _ = gpu_c1
    ot, os_ = orc.Cloud(d["tp"], d["tn"], d["tc"]), orc.Cloud(d["sp"], d["sn"], d["sc"])
    gv, ov = sga.GaussianVoxelMap(1.0), orc.VoxelMap(None, 1.0)
    gv.set_lru(2, 3)
    ov.set_lru(2, 3)
    sizes = []
    for step in range(8):
        T = se3([0.1, 0.2, 1.0], 0.02 * step, [6.0 * step, -2.0 * step, 0.1 * step])
        g, o = (tgt, ot) if step % 2 == 0 else (src, os_)
        gv.insert(g, T)
        ov.insert(o, T)
        gc, gm, g6, gn = gv.download()
        oc, om, ocv, on = ov.get()
        assert gv.size() == len(ov) and (gc == oc).all() and (gn == on).all(), step
        scale = max(1.0, float(np.abs(om).max()))
        assert np.abs(gm - om).max() <= 2e-7 * scale, step  # fp32 export of identical fp64 state
        assert np.abs(sga.api.mats_from_sym6(g6.astype(np.float64)) - ocv).max() <= 2e-7, step
        sizes.append(gv.size())
    assert min(np.diff(sizes)) < 0
    # registering against the accumulated model agrees with the oracle doing the same (VGICP, registration_helper.cpp:125-137)
    st = sga.make_setting("GICP")
    T0 = se3([0.1, 0.2, 1.0], 0.02 * 7, [42.0, -14.0, 0.7])
    res = sga.Problem(gv, src, T0).align(st, T0)
    ores = orc.align(ov, os_, orc.default_setting(factor_kind=orc.GICP, num_threads=1), T0)
    dt, dr = pose_error(res.T_target_source, ores.T_target_source)
    assert dt < POSE_TOL_T and dr < POSE_TOL_R and res.iterations == ores.iterations and res.num_inliers == ores.num_inliers, (dt, dr)


@pytest.mark.parametrize("offsets", [7, 27])
def test_gaussian_voxelmap_searched_over_7_and_27_voxels(orc, c1_f32, gpu_c1, offsets):
    """incremental_voxelmap.hpp:99-119 with set_search_offsets(7 | 27) for a GaussianVoxelMap: every voxel at the offsets offers its mean and
    the nearest wins (SURVEY a8 "7/27 selectable"; VERDICT r4 missing #2).  Linearization at two poses and the whole VGICP registration
    against the oracle with the same offsets; the one-shot map, the incremental map and the map made from host voxels agree."""
    d = c1_f32
    tgt, src, _ = gpu_c1
    ot, os_ = orc.Cloud(d["tp"], d["tn"], d["tc"]), orc.Cloud(d["sp"], d["sn"], d["sc"], tree=False)
    ov = orc.VoxelMap(ot, 1.0)
    ov.set_search_offsets(offsets)
    gv = sga.GaussianVoxelMap(1.0)
    gv.insert(tgt)
    gv.set_search_offsets(offsets)
    st = sga.make_setting("GICP")
    os_set = orc.default_setting(factor_kind=orc.GICP, num_threads=1)
    pb = sga.Problem(gv, src)
    f = orc.Factors(len(os_))
    wide = 0
    for T in (np.eye(4), se3([0.1, 0.2, 1.0], np.deg2rad(0.7), [0.49, 0.12, -0.02])):
        H, b, e, n = pb.linearize(st.factor, T)
        Ho, bo, eo, no = orc.linearize(ov, os_, os_set, T, f)
        assert abs(int(n) - int(no)) <= 3, (n, no)
        assert np.abs(H - Ho).max() <= 5e-4 * np.abs(Ho).max() and abs(e - eo) <= 5e-4 * abs(eo), (np.abs(H - Ho).max() / np.abs(Ho).max(), e, eo)
        ti, _ = pb.factors()
        oti, _ = f.get(is_voxelmap=True)
        assert (ti == oti).mean() > 0.999, (ti == oti).mean()  # voxel ids (first-insertion order on both sides), -1 = no correspondence
        one = sga.GaussianVoxelMap(1.0)
        one.insert(tgt)
        H1, _, _, n1 = sga.Problem(one, src).linearize(st.factor, T)
        wide += int(n) - int(n1)
    assert wide > 100  # the wider search does find neighbours the query's own voxel does not offer
    res = sga.Problem(gv, src).align(st)
    ores = orc.align(ov, os_, os_set)
    dt, dr = pose_error(res.T_target_source, ores.T_target_source)
    assert dt < POSE_TOL_T and dr < POSE_TOL_R and res.iterations == ores.iterations and abs(int(res.num_inliers) - int(ores.num_inliers)) <= 3, (dt, dr, res.iterations, ores.iterations)
    coords, means, c6, _ = gv.download()
    hv = sga.GaussianVoxelMap.from_voxels(1.0, coords, means, c6)
    hv.set_search_offsets(offsets)
    H2, b2, e2, n2 = sga.Problem(hv, src).linearize(st.factor, np.eye(4))
    H3, b3, e3, n3 = sga.Problem(gv, src).linearize(st.factor, np.eye(4))
    assert n2 == n3 and (H2 == H3).all() and e2 == e3


def test_voxelmap_from_host_voxels_equals_the_map_they_came_from(gpu_c1):
    """sga_index_create_voxelmap_from_voxels (the upload of a reference GaussianVoxelMap object, reduction_hip.hpp): the voxels of a device
    map, downloaded and handed back in the same order, give the same correspondences (voxel ids) and the same sums; handed back in a
    shuffled order, the same sums with permuted ids.  Inserting into such a map is refused (it has no running sums)."""
    tgt, src, _ = gpu_c1
    a = sga.GaussianVoxelMap(1.0)
    a.insert(tgt)
    coords, means, c6, _ = a.download()
    st = sga.make_setting("GICP")
    T = se3([0.1, 0.2, 1.0], 0.01, [0.2, -0.1, 0.0])
    pa = sga.Problem(a, src)
    Ha, ba, ea, na = pa.linearize(st.factor, T)
    ia, _ = pa.factors()
    b = sga.GaussianVoxelMap.from_voxels(1.0, coords, means, c6)
    assert b.size() == a.size()
    pb = sga.Problem(b, src)
    Hb, bb, eb, nb = pb.linearize(st.factor, T)
    ib, _ = pb.factors()
    assert (ia == ib).all() and na == nb and (Ha == Hb).all() and (ba == bb).all() and ea == eb  # the exported fp32 state is the state
    perm = np.random.default_rng(3).permutation(len(coords))
    c = sga.GaussianVoxelMap.from_voxels(1.0, coords[perm], means[perm], c6[perm])
    pc = sga.Problem(c, src)
    Hc, bc, ec, nc = pc.linearize(st.factor, T)
    ic, _ = pc.factors()
    assert nc == na and ((ic >= 0) == (ia >= 0)).all() and (perm[ic[ic >= 0]] == ia[ia >= 0]).all()
    assert (Hc == Ha).all() and ec == ea  # same pairs in the same source order
    with pytest.raises(sga.SgaError):
        b.insert(tgt)
    empty = sga.GaussianVoxelMap.from_voxels(1.0, np.zeros((0, 3), np.int32), np.zeros((0, 3)), np.zeros((0, 6)))
    He, be, ee, ne = sga.Problem(empty, src).linearize(st.factor, T)
    assert ne == 0 and ee == 0.0 and not He.any()


@pytest.mark.parametrize("offsets", [1, 7, 27])
def test_flat_voxelmap_from_host_voxels_equals_the_map_they_came_from(gpu_c1, offsets):
    """sga_index_create_flatmap_from_voxels (the upload of a reference IncrementalVoxelMap<FlatContainerCov>, reduction_hip.hpp): the voxels of
    a device map, downloaded and handed back, give the same packed correspondences (voxel << 32 | point) and bit-equal sums."""
    tgt, src, _ = gpu_c1
    a = sga.IncrementalVoxelMapCov(1.0)
    a.set_search_offsets(offsets)
    a.insert(tgt)
    coords, counts, pts, c6 = a.download()
    st = sga.make_setting("GICP")
    T = se3([0.1, 0.2, 1.0], 0.01, [0.2, -0.1, 0.0])
    pa = sga.Problem(a, src)
    Ha, ba, ea, na = pa.linearize(st.factor, T)
    ia, _ = pa.factors()
    b = sga.IncrementalVoxelMapCov.from_voxels(1.0, coords, counts, pts, c6, search_offsets=offsets)
    assert b.size() == a.size()
    pb = sga.Problem(b, src)
    Hb, bb, eb, nb = pb.linearize(st.factor, T)
    ib, _ = pb.factors()
    assert (ia == ib).all() and na == nb > 1000 and (Ha == Hb).all() and (ba == bb).all() and ea == eb
    assert ((ia[ia >= 0] & 0xFFFFFFFF) < 16).all() and ((ia[ia >= 0] >> 32) < a.size()).all()
    # without covariances: a target for ICP only
    c = sga.IncrementalVoxelMapCov.from_voxels(1.0, coords, counts, pts, None, search_offsets=offsets)
    Hc, bc, ec, nc = sga.Problem(c, src).linearize(sga.make_setting("ICP").factor, T)
    Hd, bd, ed, nd = sga.Problem(a, src).linearize(sga.make_setting("ICP").factor, T)
    assert nc == nd and (Hc == Hd).all() and ec == ed
    with pytest.raises(sga.SgaError):
        sga.Problem(c, src).linearize(st.factor, T)  # GICP needs the covariances
    with pytest.raises(sga.SgaError):
        b.insert(tgt)


def test_incremental_single_insert_equals_one_shot_build(gpu_c1):
    """One insert into an empty incremental map == sga_index_build_gaussian_voxelmap (the helper's one-shot path)."""
    import ctypes as C

    tgt, _, _ = gpu_c1
    inc = sga.GaussianVoxelMap(0.5)
    inc.insert(tgt)
    lib = sga._lib.load()
    h = C.c_void_p()
    sga._lib.check(lib.sga_index_build_gaussian_voxelmap(tgt.ctx.h, tgt.h, 0.5, C.byref(h)))
    try:
        one = sga.GaussianVoxelMap.__new__(sga.GaussianVoxelMap)
        one.leaf, one.ctx, one.h = 0.5, tgt.ctx, h
        a, b = inc.download(), one.download()
        assert (a[0] == b[0]).all() and (a[3] == b[3]).all()
        assert np.abs(a[1] - b[1]).max() <= 1e-6 and np.abs(a[2] - b[2]).max() <= 1e-7  # multiply-by-reciprocal vs division
    finally:
        one.h = C.c_void_p()
        lib.sga_index_destroy(h)
    # empty and degenerate inserts
    empty = sga.PointCloud(np.zeros((0, 3), np.float32), covs=np.zeros((0, 6), np.float32))
    inc.insert(empty)
    assert inc.size() == len(a[0])


def test_scan_to_model_odometry_matches_oracle(orc):
    """odometry_benchmark_small_vgicp_model_omp.cpp on the synthetic sequence: every pose of the GPU driver against the oracle
    running the same protocol (downsample -> covariances -> VGICP against the accumulated voxel map from the previous pose ->
    insert with the estimated pose)."""
    from small_gicp_amd import odometry

    frames = 6
    r = odometry.run_synthetic_model(frames)
    vm = None
    T = np.eye(4)
    for f in range(frames):
        pts, _ = sga.synthetic.kitti_like_scan(f)
        cloud = orc.Cloud(orc.voxelgrid_sampling(pts, 0.25))
        cloud.estimate_normals_covariances(20, 4)
        if vm is None:
            vm = orc.VoxelMap(None, 1.0)
            vm.insert(cloud)
        else:
            res = orc.align(vm, cloud, orc.default_setting(factor_kind=orc.GICP, num_threads=4), T)
            T = res.T_target_source
            vm.insert(cloud, T)
        dt, dr = pose_error(r["estimated"][f], T)
        assert dt < 2e-4 and dr < 2e-4, (f, dt, dr)  # the chain feeds every pose into the next map: twice the single-registration tolerance
    assert abs(r["num_voxels"] - len(vm)) <= 2


@pytest.mark.parametrize("offsets", [1, 7, 27])
def test_flat_voxelmap_matches_oracle(orc, c1_f32, gpu_c1, offsets):
    """IncrementalVoxelMap<FlatContainerCov> on the device against the oracle (pinned to the reference in
    tests/test_oracle_vs_reference.py::test_incremental_flat_voxelmap): the per-cell acceptance rule in insertion order, LRU,
    creation order, and GICP against the stored points over 1 / 7 / 27 voxels."""
    d = c1_f32
    tgt, src, _ = gpu_c1
    ot, os_ = orc.Cloud(d["tp"], d["tn"], d["tc"]), orc.Cloud(d["sp"], d["sn"], d["sc"])
    gv, ov = sga.IncrementalVoxelMapCov(1.0), orc.FlatMap(1.0)
    for m in (gv, ov):
        m.set_lru(2, 3)
        m.set_search_offsets(offsets)
    sizes = []
    for step in range(7):
        T = se3([0.1, 0.2, 1.0], 0.02 * step, [5.0 * step, -2.0 * step, 0.1 * step])
        g, o = (tgt, ot) if step % 2 == 0 else (src, os_)
        gv.insert(g, T)
        ov.insert(o, T)
        gc, gn, gp, g6 = gv.download()
        oc, on, op, ocv = ov.get()
        assert gv.size() == len(ov) and (gc == oc).all() and (gn == on).all(), step
        scale = max(1.0, float(np.abs(op).max()))
        assert np.abs(gp - op).max() <= 2e-7 * scale and np.abs(sga.api.mats_from_sym6(g6.astype(np.float64)) - ocv).max() <= 2e-7, step
        sizes.append(gv.size())
    assert min(np.diff(sizes)) < 0
    st = sga.make_setting("GICP")
    T0 = se3([0.1, 0.2, 1.0], 0.02 * 6 + 0.004, [30.0 + 0.1, -12.0 - 0.05, 0.6])
    pb = sga.Problem(gv, src, T0)
    res = pb.align(st, T0)
    ores = orc.align(ov, os_, orc.default_setting(factor_kind=orc.GICP, num_threads=1), T0)
    dt, dr = pose_error(res.T_target_source, ores.T_target_source)
    assert dt < POSE_TOL_T and dr < POSE_TOL_R and res.iterations == ores.iterations and abs(int(res.num_inliers) - int(ores.num_inliers)) <= 2, (dt, dr)
    # fixed-pose accumulators and the exported correspondences ((voxel << 32) | point)
    fac = orc.Factors(len(d["sp"]))
    oH, ob, oe, on_ = orc.linearize(ov, os_, orc.default_setting(factor_kind=orc.GICP, num_threads=1), T0, fac)
    gH, gb, ge, gn_ = pb.linearize(st.factor, T0)
    assert abs(int(gn_) - int(on_)) <= 2 and abs(ge - oe) <= 1e-4 * oe and np.abs(gH - oH).max() <= 1e-4 * np.abs(oH).max()
    gi = pb.factors()[0]
    oi = fac.get(2)[0]
    assert (gi == oi).mean() > 0.999


@pytest.mark.parametrize("offsets", [1, 7, 27])
def test_voxelmap_knn_search_matches_reference(c1_f32, gpu_c1, offsets):
    """traits::knn_search of the voxel maps, k > 1 (ann/incremental_voxelmap.hpp:127-149; VERDICT r2 missing #6): the flat map over 1 / 7 /
    27 voxels and the Gaussian map (its own voxel) against the compiled reference after the same two inserts — the same global indices
    (voxel_id << 32) | point_id in the same order, squared distances to fp32 resolution, -1 / inf beyond the number found."""
    from oracle import ref

    if not ref.available():
        pytest.skip("oracle/_ref did not travel with the repository")
    d = c1_f32
    tgt, src, _ = gpu_c1
    rt, rs = ref.Cloud(d["tp"], d["tn"], d["tc"], tree=False), ref.Cloud(d["sp"], d["sn"], d["sc"], tree=False)
    T1 = se3([0.1, 0.2, 1.0], 0.03, [0.4, -0.2, 0.05])
    gf, rf = sga.IncrementalVoxelMapCov(1.0), ref.FlatMap(1.0)
    gg, rg = sga.GaussianVoxelMap(1.0), ref.VoxelMap(1.0)
    gf.set_search_offsets(offsets)
    rf.set_search_offsets(offsets)
    for g, r in ((gf, rf), (gg, rg)):
        g.insert(tgt)
        r.insert(rt)
        g.insert(src, T1)
        r.insert(rs, T1)
    rng = np.random.default_rng(5)
    q = np.concatenate([d["sp"][rng.choice(len(d["sp"]), 1500, replace=False)].astype(np.float64) + rng.normal(0, 0.05, (1500, 3)), rng.uniform(-60, 60, (200, 3))]).astype(np.float32).astype(np.float64)
    for g, r, k in ((gf, rf, 1), (gf, rf, 6), (gf, rf, 40), (gg, rg, 1), (gg, rg, 4)):
        gi, gd = g.batch_knn_search(q, k)
        ri, rd = r.knn(q, k)
        assert gi.shape == ri.shape == (len(q), k)
        assert ((gi < 0) == (ri < 0)).all() and (np.isinf(gd) == np.isinf(rd)).all()
        ok = ri >= 0
        assert ok[:, 0].sum() > 0.5 * len(q)  # the test means something
        # distances: the stored points are the fp32 roundings of the reference's doubles
        assert np.abs(gd[ok] - rd[ok]).max() <= 1e-5 * max(1.0, float(rd[ok].max()))
        same = gi[ok] == ri[ok]
        if not same.all():  # two candidates closer to each other than fp32 resolves may swap places: then the distances agree pairwise
            rows = np.unique(np.nonzero((gi != ri) & ok)[0])
            assert len(rows) <= 0.002 * len(q), len(rows)
            for row in rows:
                assert sorted(gi[row].tolist()) == sorted(ri[row].tolist()) or np.abs(np.sort(gd[row][ok[row]]) - np.sort(rd[row][ok[row]])).max() <= 1e-5
        one_i, one_d = g.knn_search(q[0], k)
        assert (one_i == gi[0]).all()


def test_scan_to_model_gicp_odometry_matches_oracle(orc):
    """odometry_benchmark_small_gicp_model_omp.cpp (GICP against IncrementalVoxelMap<FlatContainerCov>) on the synthetic sequence."""
    from small_gicp_amd import odometry

    frames = 6
    r = odometry.run_synthetic_model(frames, model="flat")
    vm = None
    T = np.eye(4)
    for f in range(frames):
        pts, _ = sga.synthetic.kitti_like_scan(f)
        cloud = orc.Cloud(orc.voxelgrid_sampling(pts, 0.25))
        cloud.estimate_normals_covariances(20, 4)
        if vm is None:
            vm = orc.FlatMap(1.0)
            vm.insert(cloud)
        else:
            res = orc.align(vm, cloud, orc.default_setting(factor_kind=orc.GICP, num_threads=4), T)
            T = res.T_target_source
            vm.insert(cloud, T)
        dt, dr = pose_error(r["estimated"][f], T)
        assert dt < 2e-4 and dr < 2e-4, (f, dt, dr)
    assert abs(r["num_voxels"] - len(vm)) <= 2


def test_incremental_maps_edge_cases():
    """Empty and out-of-range inserts, argument checking of the scan-to-model maps."""
    pts = np.array([[0.1, 0.2, 0.3], [0.15, 0.25, 0.35], [5.0, 5.0, 5.0], [1e9, 0.0, 0.0], [np.nan, 0.0, 0.0]], np.float32)
    c6 = np.tile(np.array([1, 0, 0, 1, 0, 1], np.float32) * 0.01, (len(pts), 1))
    cloud = sga.PointCloud(pts, covs=c6)
    for cls in (sga.GaussianVoxelMap, sga.IncrementalVoxelMapCov):
        m = cls(1.0)
        assert m.size() == 0
        m.insert(sga.PointCloud(np.zeros((0, 3), np.float32), covs=np.zeros((0, 6), np.float32)))
        assert m.size() == 0
        m.insert(cloud)  # the far and the NaN point fall outside the +-2^20 cell range and are dropped
        assert m.size() == 2
        with pytest.raises(sga.SgaError):
            m.insert(sga.PointCloud(pts))  # no covariances
        with pytest.raises(sga.SgaError):
            m.set_lru(10, 0)
    g = sga.GaussianVoxelMap(1.0)
    g.set_search_offsets(7)  # (round 5: Gaussian maps are searched over 1 / 7 / 27 voxels too)
    with pytest.raises(sga.SgaError):
        g.set_search_offsets(9)
    f = sga.IncrementalVoxelMapCov(1.0)
    for bad in (0, 5, 28):
        with pytest.raises(sga.SgaError):
            f.set_search_offsets(bad)
    with pytest.raises(sga.SgaError):
        f.set_setting(0.01, 17)
    f.set_setting(0.0, 1)  # one point per cell
    f.insert(cloud)
    assert f.download()[1].tolist() == [1, 1]
    with pytest.raises(sga.SgaError):
        f.set_setting(0.01, 10)  # after the first insert
    # a voxel that receives more than max points keeps the first ones, at least min distance apart
    rng = np.random.default_rng(0)
    dense = rng.uniform(0.0, 1.0, (500, 3)).astype(np.float32)
    h = sga.IncrementalVoxelMapCov(1.0)
    h.insert(sga.PointCloud(dense, covs=np.tile(c6[:1], (500, 1))))
    coords, counts, p, _ = h.download()
    assert len(coords) == 1 and counts[0] == 10 and (p[0] == dense[0]).all()
    dmin = min(np.linalg.norm(p[i] - p[j]) for i in range(10) for j in range(i))
    assert dmin >= 0.1 - 1e-6


def test_problem_from_source_index_equals_problem_from_cloud(gpu_c1):
    """sga_problem_create_from_index takes the source in the kd order of its own index instead of sorting it by target leaf: the
    registration, the per-point factor state (reported in the cloud's original order) and the sums must be those of the sorted form."""
    tgt, src, tree = gpu_c1
    src_tree = sga.KdTree(src)  # carries the cloud's covariances / normals in kd order
    for name in ("GICP", "PLANE_ICP"):
        for mode, tol in (("fp64", 1e-9), ("fp32", 1e-5)):
            st = sga.make_setting(name, math_mode=mode)
            a, b = sga.Problem(tree, src), sga.Problem(tree, src_tree)
            T = POSES[1]
            Ha, ba, ea, na = a.linearize(st.factor, T)
            Hb, bb, eb, nb = b.linearize(st.factor, T)
            assert na == nb and (a.factors()[0] == b.factors()[0]).all()
            rel = 1e-12 if mode == "fp64" else 2e-5
            assert np.abs(Ha - Hb).max() <= rel * np.abs(Ha).max() and abs(ea - eb) <= rel * abs(ea)
            ra, rb = a.align(st), b.align(st)
            dt, dr = pose_error(ra.T_target_source, rb.T_target_source)
            assert dt < tol and dr < tol and ra.iterations == rb.iterations and ra.num_inliers == rb.num_inliers, (name, mode, dt, dr)
