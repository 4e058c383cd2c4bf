"""CPU-side checks of the product boundary (no GPU compute): the C-ABI library loads, exports every symbol include/*.h declares,
fails LOUDLY without a GPU (no CPU fallback), and its host logic — the LM/GN optimizer of registration/optimizer.hpp, se3_exp,
the accumulator layout — reproduces the oracle when driven through sga_optimize with oracle reductions as callbacks."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import small_gicp_amd as sga
from conftest import ROOT, pose_error
from small_gicp_amd import _lib


def test_library_exports_every_declared_symbol():
    assert os.path.exists(sga.LIB_PATH), "run `make lib` / __graft_entry__.build() first"
    header = open(os.path.join(ROOT, "include", "small_gicp_amd.h")).read() + open(os.path.join(ROOT, "include", "small_gicp_amd_debug.h")).read()
    declared = set(re.findall(r"\b(sga_[a-z0-9_]+)\s*\(", header))
    typedefs = {"sga_linearize_fn", "sga_error_fn"}
    declared -= typedefs
    lib = C.CDLL(sga.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert declared == bound, (declared - bound, bound - declared)


def test_no_gpu_fails_loudly():
    lib = sga.load()
    if lib.sga_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(sga.SgaError) as ei:
        sga.Context(0)
    assert "no CPU fallback" in str(ei.value) or "error 3" in str(ei.value)
    with pytest.raises(sga.SgaError):
        sga.PointCloud(np.zeros((10, 3), np.float32))


def test_argument_checks_come_before_any_device_work():
    """Entry points that take host arrays reject null handles and impossible sizes with SGA_ERR_INVALID before they touch a device (so
    the checks run on a box without a GPU): the from-host-voxels creators and the sga_multi setters added in round 4."""
    lib = sga.load()
    out = C.c_void_p()
    one3 = (C.c_int32 * 3)(0, 0, 0)
    cnt = (C.c_uint32 * 1)(1)
    d = (C.c_double * 96)()
    f = (C.c_float * 6)()
    INVALID = 1  # SGA_ERR_INVALID
    assert lib.sga_index_create_voxelmap_from_voxels(None, 1.0, one3, d, d, 1, C.byref(out)) == INVALID
    assert lib.sga_index_create_flatmap_from_voxels(None, 1.0, one3, cnt, d, d, 1, 1, C.byref(out)) == INVALID
    assert lib.sga_multi_set_target_voxels(None, 1.0, one3, d, d, 1) == INVALID
    assert lib.sga_multi_set_target_flat_voxels(None, 1.0, one3, cnt, d, d, 1, 1) == INVALID
    assert lib.sga_multi_set_target_f32(None, f, None, None, 1) == INVALID
    assert lib.sga_multi_set_source_f32(None, f, None, None, 1, d) == INVALID
    assert lib.sga_multi_create(None, 1, C.byref(out)) == INVALID and lib.sga_multi_create((C.c_int * 1)(0), 0, C.byref(out)) == INVALID
    # round 5: origins, clones, rejectors, search offsets
    assert lib.sga_cloud_create_f32_origin(None, f, None, None, 1, d, C.byref(out)) == INVALID
    assert lib.sga_cloud_create_f64_origin(None, d, None, None, 1, d, C.byref(out)) == INVALID
    assert lib.sga_cloud_origin(None, d) == INVALID and lib.sga_index_origin(None, d) == INVALID
    assert lib.sga_cloud_download_f64(None, None, d, None, None) == INVALID
    assert lib.sga_index_clone(None, None, C.byref(out)) == INVALID
    assert lib.sga_multi_set_target_f32_origin(None, f, None, None, 1, d) == INVALID
    assert lib.sga_multi_set_source_f32_origin(None, f, None, None, 1, d, d) == INVALID
    assert lib.sga_multi_set_rejector(None, None, None) == INVALID and lib.sga_multi_set_search_offsets(None, 7) == INVALID
    assert lib.sga_voxelmap_set_search_offsets(None, 7) == INVALID
    assert b"" != lib.sga_last_error()


def test_origin_rule_and_frame_conversions_on_the_host():
    """sga_choose_origin (small_gicp_amd.h, device frames): the bounding-box centre rounded to a multiple of 128 m — 0 for everything centred
    within 64 m of the origin (so the BASELINE configs are stored as before), 0 for empty / non-finite boxes."""
    lib = sga.load()

    def origin(lo, hi):
        o = (C.c_double * 3)()
        lib.sga_choose_origin((C.c_double * 3)(*lo), (C.c_double * 3)(*hi), o)
        return list(o)

    assert origin([-50, -50, -1], [50, 50, 10]) == [0.0, 0.0, 0.0]
    assert origin([10, -100, 0], [117.9, 27.9, 5]) == [0.0, 0.0, 0.0]      # centres (63.95, -36.05, 2.5)
    assert origin([10, -100, 0], [118.1, -28.1, 5]) == [128.0, -128.0, 0.0]  # centres (64.05, -64.05, 2.5)
    assert origin([99990, 199990, 290], [100040, 200080, 310]) == [99968.0, 200064.0, 256.0]  # centres (100015, 200035, 300)
    assert origin([1, 1, 1], [-1, -1, -1]) == [0.0, 0.0, 0.0]
    assert origin([float("inf")] * 3, [float("-inf")] * 3) == [0.0, 0.0, 0.0]
    assert origin([float("nan"), 0, 0], [1, 1, 1])[0] == 0.0


def test_defaults_match_reference():
    """registration_helper.hpp:37-49, optimizer.hpp:66,83, termination_criteria.hpp:13, rejector.hpp:20."""
    s = _lib.RegistrationSettingC()
    sga.load().sga_registration_setting_default(C.byref(s))
    assert s.factor.factor_kind == sga.GICP and s.factor.max_dist_sq == 1.0 and s.factor.robust_kind == 0 and s.factor.robust_c == 1.0
    assert (s.optimizer, s.max_iterations, s.max_inner_iterations) == (0, 20, 10)
    assert (s.init_lambda, s.lambda_factor, s.gn_lambda) == (1e-3, 10.0, 1e-6)
    assert s.translation_eps == 1e-3 and abs(s.rotation_eps - 0.1 * np.pi / 180.0) < 1e-18


def test_se3_exp_matches_oracle(orc):
    rng = np.random.default_rng(0)
    for scale in (1e-9, 1e-4, 0.2, 1.5):
        a = np.ascontiguousarray(rng.normal(size=6) * scale)
        T = np.empty(16)
        sga.load().sga_se3_exp(a.ctypes.data_as(C.POINTER(C.c_double)), T.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.allclose(T.reshape(4, 4).T, orc.se3_exp(a), atol=1e-14)


def test_unpack_accumulator_layout():
    acc = np.arange(30, dtype=np.float64)
    acc[28] = 1234.0
    H, b, e, n = sga.unpack_accumulator(acc)
    assert np.allclose(H, H.T)
    k = 0
    for i in range(6):
        for j in range(i, 6):
            assert H[i, j] == k
            k += 1
    assert (b == np.arange(21, 27)).all() and e == 27 and n == 1234


@pytest.mark.parametrize("case", ["GICP", "PLANE_ICP", "ICP", "GN_GICP"])
def test_host_optimizer_reproduces_oracle(orc, c1_oracle_clouds, case):
    """sga_optimize (product host code) driven by oracle reductions == the oracle's own optimizer: same pose, iteration count,
    inliers, H, b, error.  Exercises LM accept/reject, lambda schedule, termination and result bookkeeping on the CPU."""
    tc, sc = c1_oracle_clouds
    kind = {"GICP": orc.GICP, "PLANE_ICP": orc.PLANE_ICP, "ICP": orc.ICP, "GN_GICP": orc.GICP}[case]
    opt = 1 if case.startswith("GN") else 0
    s = orc.default_setting(factor_kind=kind, num_threads=1, optimizer_type=opt)
    ref = orc.align(tc, sc, s)
    f = orc.Factors(len(sc))

    def lin(T):
        return orc.linearize(tc, sc, s, T, f)

    def err(T):
        return orc.error(tc, sc, s, T, f)

    st = sga.make_setting({orc.GICP: "GICP", orc.PLANE_ICP: "PLANE_ICP", orc.ICP: "ICP"}[kind], optimizer="GN" if opt else "LM")
    res = sga.optimize(st, np.eye(4), lin, err)
    assert res.iterations == ref.iterations and res.converged == ref.converged and res.num_inliers == ref.num_inliers
    dt, dr = pose_error(res.T_target_source, ref.T_target_source)
    assert dt < 1e-10 and dr < 1e-7
    assert np.allclose(res.H, ref.H, rtol=1e-9) and np.allclose(res.b, ref.b, rtol=1e-6, atol=1e-6) and abs(res.error - ref.error) <= 1e-9 * abs(ref.error)


def test_host_optimizer_lm_failure_path():
    """optimizer.hpp:141-143: 10 rejected trials end the optimisation with converged = False."""
    calls = {"lin": 0, "err": 0}

    def lin(T):
        calls["lin"] += 1
        return np.eye(6), np.ones(6), 1.0, 5

    def err(T):
        calls["err"] += 1
        return 2.0  # never better

    res = sga.optimize(sga.make_setting("GICP"), np.eye(4), lin, err)
    assert not res.converged and res.iterations == 0 and calls == {"lin": 1, "err": 10}
    assert np.allclose(res.T_target_source, np.eye(4))


def test_restrict_dof_factor():
    """general_factor.hpp:41-75: H += lambda * diag(|mask - 1|) freezes the masked twist components."""
    rng = np.random.default_rng(0)
    J = rng.normal(size=(40, 6))
    H0, b0 = J.T @ J, rng.normal(size=6)
    seen = {}

    def lin(T):
        return H0, b0, 10.0, 40

    def err(T):
        seen["T"] = T
        return 0.0

    mask = [1, 1, 0, 1, 1, 0]  # freeze yaw and z
    st = sga.make_setting("GICP", max_iterations=1, restrict_dof_lambda=1e9, restrict_dof_mask=mask)
    res = sga.optimize(st, np.eye(4), lin, err)
    Hc = H0 + 1e9 * np.diag(np.abs(np.array(mask, dtype=np.float64) - 1.0))
    delta = np.linalg.solve(Hc + 1e-3 * np.eye(6), -b0)
    assert abs(delta[2]) < 1e-8 and abs(delta[5]) < 1e-8
    from oracle import orc

    assert np.allclose(res.T_target_source, orc.se3_exp(delta), atol=1e-9)
    assert res.H[2, 2] > 1e8 and res.H[5, 5] > 1e8  # result.H carries the general factor like the reference


def test_callback_error_propagates():
    def lin(T):
        raise ValueError("boom")

    with pytest.raises(ValueError):
        sga.optimize(sga.make_setting("GICP"), np.eye(4), lin, lambda T: 0.0)


def test_small_gicp_module_name_exports_the_reference_surface():
    """`import small_gicp` resolves to this repository's drop-in package and lists the names the reference's Python tests use
    (no device call is made by importing it)."""
    import inspect

    import small_gicp

    for name in ("PointCloud", "KdTree", "GaussianVoxelMap", "RegistrationResult", "read_ply", "voxelgrid_sampling", "estimate_normals", "estimate_covariances",
                 "estimate_normals_covariances", "preprocess_points", "align", "DistanceRejector", "ICPFactor", "PointToPlaneICPFactor", "GICPFactor"):
        assert hasattr(small_gicp, name), name
    # keyword names and defaults of the binding (src/python/align.cpp:95-106, preprocess.cpp:233-236)
    p = inspect.signature(small_gicp._align_points).parameters
    assert list(p)[:5] == ["target_points", "source_points", "init_T_target_source", "registration_type", "voxel_resolution"]
    assert p["downsampling_resolution"].default == 0.25 and p["max_iterations"].default == 20 and p["translation_epsilon"].default == 1e-3
    q = inspect.signature(small_gicp.preprocess_points).parameters
    assert q["num_neighbors"].default == 10 and q["downsampling_resolution"].default == 0.25
    assert small_gicp.DistanceRejector().max_dist_sq == 1.0


def test_shard_frame_check_is_exact_for_any_rank_count():
    """the ranks of a sharded registration compare their source origins through ONE sum over the ranks
    (linearize.hip: problem_check_shard_frames).  The sums must be exact whatever the origin and the number of ranks: identical
    origins are accepted (n = 1 .. 9, origins up to 1e6 m with fractional parts), a single deviating rank — by one ulp — is refused,
    and every rank takes the same decision (the decision is a function of the summed values alone).  The sums are integers below 2^53:
    their value does not depend on the order in which a transport adds them (checked by summing in shuffled orders)."""
    lib = sga.load()
    N = 32

    def pack(o):
        o = np.ascontiguousarray(o, dtype=np.float64)
        out = np.zeros(N)
        lib.sga_debug_shard_frame_pack(o.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def agree(total):
        total = np.ascontiguousarray(total, dtype=np.float64)
        return bool(lib.sga_debug_shard_frame_agree(total.ctypes.data_as(C.POINTER(C.c_double))))

    rng = np.random.default_rng(7)
    for n in range(1, 10):
        for _ in range(200):
            o = rng.uniform(-1e6, 1e6, 3)
            if rng.random() < 0.2:
                o = np.round(o / 128.0) * 128.0  # the library's own origins
            rows = [pack(o) for _ in range(n)]
            for _ in range(3):  # any order of the additions gives the same doubles
                rng.shuffle(rows)
                tot = np.zeros(N)
                for r in rows:
                    tot = tot + r
                assert agree(tot), (n, o)
            if n >= 2:
                bad = o.copy()
                k = rng.integers(3)
                bad[k] = np.nextafter(bad[k], np.inf) if rng.random() < 0.5 else bad[k] + 128.0 * rng.integers(1, 5)
                rows[rng.integers(n)] = pack(bad)
                assert not agree(np.sum(rows, axis=0)), (n, o, bad)
    assert agree(pack([0.0, -0.0, 0.0]) + pack([-0.0, 0.0, 0.0]))  # one origin
    # 1024 ranks at the largest piece value: still exact
    big = pack([-np.nextafter(0.0, 1.0)] * 3)  # (bit pattern with high pieces set)
    assert agree(big * 1024)
