"""The CPU restatement (oracle/) against the REFERENCE ITSELF: oracle/_ref/libsmall_gicp_ref.so is the unmodified koide3/small_gicp
code (headers + registration_helper.cpp) compiled in place from /root/reference over a home-made Eigen stand-in
(oracle/ref/eigen_shim: the reference needs Eigen, which this image does not ship).  Every stage of the hot path must agree to
rounding: voxel grid, kd-tree kNN, normals/covariances, per-factor linearization incl. robust kernels, the error pass, and the
full registrations of registration_helper.cpp (ICP / PLANE_ICP / GICP / VGICP, serial and OpenMP reductions).
Skipped where the library cannot be built (no /root/reference, e.g. on the GPU box unless the built .so travelled with the repo)."""
import numpy as np
import pytest

from conftest import pose_error
from oracle import ref

pytestmark = pytest.mark.skipif(not (ref.available() or ref.build()), reason="oracle/_ref not built (needs /root/reference)")


@pytest.fixture(scope="module")
def clouds(orc, c1_raw):
    tgt, src, _ = c1_raw
    rt_raw, rs_raw = ref.Cloud(tgt, tree=False), ref.Cloud(src, tree=False)
    rt, rs = rt_raw.voxelgrid_sampling(0.25), rs_raw.voxelgrid_sampling(0.25)
    rt.estimate_normals_covariances(10, 1)
    rs.estimate_normals_covariances(10, 1)
    td, sd = orc.voxelgrid_sampling(tgt, 0.25), orc.voxelgrid_sampling(src, 0.25)
    ot, os_ = orc.Cloud(td), orc.Cloud(sd)
    ot.estimate_normals_covariances(10, 1)
    os_.estimate_normals_covariances(10, 1)
    return dict(rt=rt, rs=rs, ot=ot, os=os_)


def test_voxelgrid_and_features(clouds):
    for r, o in ((clouds["rt"], clouds["ot"]), (clouds["rs"], clouds["os"])):
        rp, rn, rc = r.get()
        op, on, oc = o.get()
        assert rp.shape == op.shape
        assert np.abs(rp - op).max() < 1e-10  # same voxels in the same (ascending key) order
        assert np.abs(rn - on).max() < 1e-6 and np.abs(rc - oc).max() < 1e-6  # eigenvectors of near-degenerate neighbourhoods amplify summation-order rounding


def test_knn(clouds):
    rp = clouds["rt"].get()[0]
    rng = np.random.default_rng(1)
    q = np.concatenate([rp[rng.choice(len(rp), 60)], rp[rng.choice(len(rp), 60)] + rng.normal(0, 1, (60, 3)), rng.uniform(0, 100, (60, 3))])
    ri, rd = clouds["rt"].knn(q, 20)
    oi, od = clouds["ot"].knn(q, 20)
    assert (ri == oi).all() and (np.abs(rd - od) <= 1e-13 * np.maximum(1.0, od)).all()


@pytest.mark.parametrize("type_,robust", [(0, 0), (1, 0), (2, 0), (2, 1), (2, 2)])
@pytest.mark.parametrize("threads", [1, 4])
def test_linearize_and_error(orc, clouds, type_, robust, threads):
    T = np.eye(4)
    T[:3, 3] = [0.3, 0.1, -0.02]
    c, s = np.cos(0.01), np.sin(0.01)
    T[:3, :3] = [[c, -s, 0], [s, c, 0], [0, 0, 1]]
    for pose in (np.eye(4), T):
        H, b, e, e2, n = ref.linearize(clouds["rt"], clouds["rs"], type_, robust, 0.7, 1.0, threads, pose)
        st = orc.default_setting(factor_kind=type_, robust_kind=robust, robust_c=0.7, num_threads=threads)
        f = orc.Factors(len(clouds["os"]))
        Ho, bo, eo, no = orc.linearize(clouds["ot"], clouds["os"], st, pose, f)
        assert n == no
        assert np.abs(H - Ho).max() <= 1e-8 * np.abs(H).max()  # the two sides estimate their own covariances (equal to ~1e-8)
        assert np.abs(b - bo).max() <= 1e-7 * np.sqrt(np.abs(np.diag(H)).max() * e)
        assert abs(e - eo) <= 1e-8 * e
        assert abs(e2 - orc.error(clouds["ot"], clouds["os"], st, pose, f)) <= 1e-8 * e


@pytest.mark.parametrize("type_", [0, 1, 2, 3])
@pytest.mark.parametrize("threads", [1, 4])
def test_align(orc, clouds, c1_raw, type_, threads):
    r = ref.align(clouds["rt"], clouds["rs"], type_, 1.0, 1.0, threads)
    st = orc.default_setting(factor_kind=min(type_, 2), num_threads=threads)
    target = orc.VoxelMap(clouds["ot"], 1.0) if type_ == 3 else clouds["ot"]
    o = orc.align(target, clouds["os"], st)
    assert (r.iterations, r.converged, r.num_inliers) == (o.iterations, o.converged, o.num_inliers)
    dt, dr = pose_error(r.T_target_source, o.T_target_source)
    assert dt < 1e-9 and dr < 1e-7, (dt, dr)
    assert abs(r.error - o.error) <= 1e-9 * abs(o.error) and np.abs(r.H - o.H).max() <= 1e-9 * np.abs(o.H).max()
    # and the reference itself meets its own test tolerance on this pair (src/test/helper_test.cpp:27-39)
    dt, dr = pose_error(r.T_target_source, c1_raw[2])
    assert dt < 0.2 and dr < np.deg2rad(2.5)
    if type_ == 3:
        assert clouds["rt"].voxelmap_size(1.0) == len(target)


def test_goldens_equal_reference(clouds, c1_gold):
    """tests/golden/c1_oracle.json (the 1e-4 anchor of the GPU tests) is what the reference computes."""
    for name, t in (("ICP", 0), ("PLANE_ICP", 1), ("GICP", 2), ("VGICP", 3)):
        r = ref.align(clouds["rt"], clouds["rs"], t, 1.0, 1.0, 1)
        g = c1_gold["cases"][name]
        dt, dr = pose_error(r.T_target_source, np.array(g["T"]))
        assert dt < 1e-9 and dr < 1e-7 and r.iterations == g["iterations"] and r.num_inliers == g["num_inliers"]


def _se3(axis, ang, t):
    from scipy.spatial.transform import Rotation

    T = np.eye(4)
    T[:3, :3] = Rotation.from_rotvec(np.asarray(axis, dtype=np.float64) / np.linalg.norm(axis) * ang).as_matrix()
    T[:3, 3] = t
    return T


def test_incremental_voxelmap(orc, clouds):
    """Scan-to-model target: the same sequence of insert(cloud, T) into the reference's GaussianVoxelMap and into the oracle's —
    creation order of the voxels, running means / mean covariances, point counts, and the LRU sweep (small horizon and cycle so
    that voxels really get removed) — incremental_voxelmap.hpp:55-92, gaussian_voxelmap.hpp:32-53."""
    rv, ov = ref.VoxelMap(1.0), orc.VoxelMap(None, 1.0)
    rv.set_lru(2, 3)
    ov.set_lru(2, 3)
    # identical inputs on both sides (the oracle's points and covariances), so that only the map arithmetic is compared
    pairs = []
    for o in (clouds["ot"], clouds["os"]):
        p, n, c = o.get()
        pairs.append((ref.Cloud(p, n, c, tree=False), o))
    sizes = []
    for step in range(8):
        # a "sensor" drifting away: later scans stop touching the first voxels, which the sweep then removes
        T = _se3([0.1, 0.2, 1.0], 0.02 * step, [6.0 * step, -2.0 * step, 0.1 * step])
        r, o = pairs[step % 2]
        rv.insert(r, T)
        ov.insert(o, T)
        rc, rm, rcv, rn = rv.get()
        oc, om, ocv, on = ov.get()
        assert len(rv) == len(ov) and (rc == oc).all() and (rn == on).all(), step
        assert np.abs(rm - om).max() < 1e-12 and np.abs(rcv - ocv).max() < 1e-13, step
        sizes.append(len(ov))
    assert min(np.diff(sizes)) < 0  # the sweep did remove voxels at some step


@pytest.mark.parametrize("offsets", [1, 7, 27])
def test_incremental_flat_voxelmap(orc, clouds, offsets):
    """IncrementalVoxelMap<FlatContainerCov> (scan-to-model GICP target): insert sequence with poses and LRU, the per-cell
    acceptance rule (max points, minimum distance), the offset patterns of the search, and a registration against the map
    (flat_container.hpp:33-93, incremental_voxelmap.hpp:55-190, odometry_benchmark_small_gicp_model_omp.cpp:33-36)."""
    rv, ov = ref.FlatMap(1.0), orc.FlatMap(1.0)
    for m in (rv, ov):
        m.set_lru(2, 3)
        m.set_search_offsets(offsets)
    pairs = []
    for o in (clouds["ot"], clouds["os"]):
        p, n, c = o.get()
        pairs.append((ref.Cloud(p, n, c, tree=False), o))
    sizes = []
    for step in range(7):
        T = _se3([0.1, 0.2, 1.0], 0.02 * step, [5.0 * step, -2.0 * step, 0.1 * step])
        r, o = pairs[step % 2]
        rv.insert(r, T)
        ov.insert(o, T)
        rc, rn, rp, rcv = rv.get()
        oc, on, op, ocv = ov.get()
        assert len(rv) == len(ov) and (rc == oc).all() and (rn == on).all(), step
        assert rn.max() <= 10 and np.abs(rp - op).max() < 1e-12 and np.abs(rcv - ocv).max() < 1e-13, step
        sizes.append(len(ov))
    assert min(np.diff(sizes)) < 0
    # register the other cloud against the accumulated model from a nearby pose
    T0 = _se3([0.1, 0.2, 1.0], 0.02 * 6 + 0.004, [30.0 + 0.1, -12.0 - 0.05, 0.6])
    rr = rv.align(pairs[1][0], T0, num_threads=1)
    orr = orc.align(ov, pairs[1][1], orc.default_setting(factor_kind=orc.GICP, num_threads=1), T0)
    dt, dr = pose_error(rr.T_target_source, orr.T_target_source)
    assert dt < 1e-8 and dr < 1e-8 and rr.iterations == orr.iterations and rr.num_inliers == orr.num_inliers
    assert abs(rr.error - orr.error) <= 1e-8 * abs(rr.error)


@pytest.mark.parametrize("optimizer", [0, 1])
@pytest.mark.parametrize("restrict", [False, True])
def test_general_factor_and_optimizers_match_reference(orc, clouds, optimizer, restrict):
    """RestrictDoFFactor (factors/general_factor.hpp:41-75) and both optimizers (optimizer.hpp:24-149): the restatement equals
    Registration<GICPFactor, ParallelReductionOMP, RestrictDoFFactor | NullFactor, DistanceRejector, LM | GN> of the reference."""
    rt, rs, ot, os_ = clouds["rt"], clouds["rs"], clouds["ot"], clouds["os"]
    mask = (0.0, 0.0, 1.0, 1.0, 1.0, 0.0)  # yaw + planar translation free; roll, pitch, z softly frozen
    lam = 1e9 if restrict else 0.0
    r = ref.align_general(rt, rs, optimizer, lam, mask, 1.0, 1)
    s = orc.default_setting(factor_kind=orc.GICP, num_threads=1, optimizer_type=optimizer, restrict_lambda=lam, restrict_mask=mask)
    o = orc.align(ot, os_, s)
    assert o.iterations == r.iterations and o.num_inliers == r.num_inliers and o.converged == r.converged
    assert np.abs(o.T_target_source - r.T_target_source).max() < 1e-9
    assert np.abs(o.H - r.H).max() <= 1e-9 * np.abs(r.H).max()
    if restrict:  # the frozen directions really are frozen (softly: lambda = 1e9)
        E = r.T_target_source
        assert abs(E[2, 3]) < 1e-3 and abs(E[2, 0]) < 1e-3 and abs(E[2, 1]) < 1e-3
