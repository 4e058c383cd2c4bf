"""Warm linearization passes (certified neighbours, linearize.hip) give the same correspondences and the same sums as cold passes
(the full walk for every source point) — on a real MI355X through the C-ABI.

A warm pass keeps the exact neighbour of the previous linearization wherever its certificate holds (new distance < exclusion
radius - motion) and walks the tree only for the other points.  Exactness is by construction; these tests check it on whole pose
chains shaped like a registration (large steps first, then ever smaller ones), on tie-heavy lattices, with the rejector on and
off, for every factor, and with the limit forced so that nearly every certificate fails.
"""
import os

import numpy as np
import pytest

import small_gicp_amd as sga
from conftest import pose_error

pytestmark = pytest.mark.gpu


def se3(axis, ang, t):
    from scipy.spatial.transform import Rotation

    T = np.eye(4)
    T[:3, :3] = Rotation.from_rotvec(np.asarray(axis, dtype=np.float64) / np.linalg.norm(axis) * ang).as_matrix()
    T[:3, 3] = t
    return T


def pose_chain(T_goal, fractions=(0.0, 0.55, 0.9, 0.985, 0.998, 0.9997, 0.99996, 1.0, 1.0)):
    """Poses approaching T_goal like an LM run: screw interpolation at the given fractions (steps shrink by ~an order each)."""
    from scipy.spatial.transform import Rotation

    rv = Rotation.from_matrix(T_goal[:3, :3]).as_rotvec()
    out = []
    for f in fractions:
        T = np.eye(4)
        T[:3, :3] = Rotation.from_rotvec(rv * f).as_matrix()
        T[:3, 3] = T_goal[:3, 3] * f
        out.append(T)
    return out


@pytest.fixture(autouse=True)
def restore_limits():
    lim = sga.get_warm_limit()
    yield
    sga.set_warm_limit(lim)


def run_chain(tree, src, setting, poses, limit, rel):
    """linearize along `poses` on two problems: one with warm passes (limits), one cold only; everything must agree."""
    pw, pc = sga.Problem(tree, src), sga.Problem(tree, src)
    for k, T in enumerate(poses):
        sga.set_warm_limit(limit)
        Hw, bw, ew, nw = pw.linearize(setting.factor, T)
        e2w = pw.error(setting.factor, T)
        cw, mw = pw.factors()
        sga.set_warm_limit(-1.0)
        Hc, bc, ec, nc = pc.linearize(setting.factor, T)
        cc, mc = pc.factors()
        assert (cw == cc).all(), (k, int((cw != cc).sum()))
        assert nw == nc, (k, nw, nc)
        scale = max(np.abs(Hc).max(), 1e-30)
        assert np.abs(Hw - Hc).max() <= rel * scale and np.abs(bw - bc).max() <= rel * max(np.abs(bc).max(), scale * 1e-3), k
        assert abs(ew - ec) <= rel * max(abs(ec), 1e-30) and abs(e2w - ew) <= 1e-5 * max(abs(ew), 1e-30), k
        assert np.abs(mw - mc).max(axis=1).max() <= 0 or (np.abs(mw - mc).max(axis=1) <= 1e-4 * np.abs(mc).max(axis=1)).all(), k  # the cached mahalanobis: same pairs (the two kernels may round the 3x3 inverse differently)
    sw, sc = pw.pass_stats(), pc.pass_stats()
    assert sc["warm_passes"] == 0 and sc["cold_passes"] == len(poses)
    return sw


@pytest.mark.parametrize("kind", ["GICP", "PLANE_ICP", "ICP"])
@pytest.mark.parametrize("mode", ["fp32", "fp64"])
def test_warm_equals_cold_on_c1(c1_f32, kind, mode):
    d = c1_f32
    tgt = sga.PointCloud(d["tp"], d["tn"], d["tc"])
    src = sga.PointCloud(d["sp"], d["sn"], d["sc"])
    tree = sga.KdTree(tgt)
    st = sga.make_setting(kind, math_mode=mode)
    goal = se3([0.1, 0.2, 1.0], np.deg2rad(0.7), [0.49, 0.12, -0.02])
    stats = run_chain(tree, src, st, pose_chain(goal), sga.get_warm_limit(), 1e-6 if mode == "fp32" else 1e-12)
    assert stats["warm_passes"] >= 4, stats  # the small steps of the chain
    assert stats["walked_points"] < 0.3 * len(d["sp"]) * stats["warm_passes"], stats  # and most of their points kept their neighbour without a walk


def test_warm_with_every_certificate_failing(c1_f32):
    """Limit forced wide open: every pass after the first is 'warm' although the points move by decimetres, so nearly every
    certificate fails and nearly every point walks — still the same answer."""
    d = c1_f32
    tgt = sga.PointCloud(d["tp"], d["tn"], d["tc"])
    src = sga.PointCloud(d["sp"], d["sn"], d["sc"])
    tree = sga.KdTree(tgt)
    goal = se3([0.3, -0.2, 1.0], np.deg2rad(3.0), [0.6, -0.3, 0.1])
    for st in (sga.make_setting("GICP"), sga.make_setting("GICP", max_correspondence_distance=None), sga.make_setting("ICP", max_correspondence_distance=0.3)):
        stats = run_chain(tree, src, st, pose_chain(goal, (0.0, 0.3, 0.6, 0.8, 0.9, 1.0)), 100.0, 1e-6)
        assert stats["warm_passes"] == 5 and stats["walked_points"] > 0.5 * len(d["sp"]), stats


def test_warm_on_lattice_ties():
    """Integer-lattice target (every query has equidistant candidates, kdtree_synthetic_test.cpp:26-76 style) and sub-millimetre
    steps: the canonical tie rule (lowest kd position) makes warm and cold agree on every correspondence."""
    g = np.arange(-6, 7, dtype=np.float32)
    lattice = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    rng = np.random.default_rng(3)
    src_pts = np.concatenate([lattice[rng.choice(len(lattice), 800, replace=False)] + 0.5, lattice[:500] + np.float32(0.25), rng.uniform(-6, 6, (700, 3)).astype(np.float32)]).astype(np.float32)
    tree = sga.KdTree(sga.PointCloud(lattice))
    src = sga.PointCloud(src_pts)
    st = sga.make_setting("ICP", max_correspondence_distance=2.0)
    poses = [np.eye(4)] + [se3([0, 0, 1], 1e-5 * k, [2e-4 * k, -1e-4 * k, 0.0]) for k in range(1, 6)] + [np.eye(4)]
    stats = run_chain(tree, src, st, poses, 0.1, 1e-6)
    assert stats["warm_passes"] == len(poses) - 1


def test_warm_registration_matches_cold_registration_100k():
    """Whole registrations (LM from the identity to convergence) on the synthetic 100k pair: identical iteration count, inliers and
    final correspondences; poses equal to rounding of the fp64 sums."""
    target, source, T_gt = sga.synthetic.registration_pair(100_000)
    tgt, src = sga.PointCloud(target), sga.PointCloud(source)
    sga.estimate_covariances(tgt, None, 20)
    sga.estimate_covariances(src, None, 20)
    tree = sga.KdTree(tgt)
    st = sga.make_setting("GICP", max_correspondence_distance=1.0, translation_eps=1e-7, rotation_eps=1e-8)  # iterate on into the small steps
    pw, pc = sga.Problem(tree, src), sga.Problem(tree, src)
    rw = pw.align(st)
    sga.set_warm_limit(-1.0)
    rc = pc.align(st)
    assert rw.iterations == rc.iterations and rw.num_inliers == rc.num_inliers and rw.converged == rc.converged
    dt, dr = pose_error(rw.T_target_source, rc.T_target_source)
    assert dt < 1e-9 and dr < 1e-9, (dt, dr)
    assert (pw.factors()[0] == pc.factors()[0]).all()
    sw = pw.pass_stats()
    assert sw["warm_passes"] >= 1 and sw["cold_passes"] >= 1, sw
    dt, dr = pose_error(rw.T_target_source, T_gt)
    assert dt < 2e-2 and dr < 2e-3


def test_streaming_warm_kernel_walks_the_tree_on_a_deep_index():
    """certify_linearize_kernel (>= 262144 source points, millimetre motions) with its walkers sent through the kd walk (no cell grid:
    SGA_GRID = 0) over a 1M-point target — a tree of depth 17, where each of the workgroup's four waves needs its own full-depth
    traversal stack (the stacks were indexed by threadIdx.x instead of the lane, so wave w worked w rows further down and
    wave 3 could reach past the allocation).  Warm == cold on a chain of millimetre steps, rejector tight and wide."""
    target, source, T_gt = sga.synthetic.registration_pair(1_000_000)
    sga.set_grid_mode(0)
    try:
        tree = sga.KdTree(sga.PointCloud(target))
    finally:
        sga.set_grid_mode(1)
    src = sga.PointCloud(source[:300_000])
    steps = [np.eye(4)] + [se3([0.3, -0.2, 1.0], 2e-6 * k, [4e-4 * k, -3e-4 * k, 2e-4 * k]) for k in range(1, 6)]
    poses = [T_gt @ S for S in steps]
    limit = sga.get_warm_limit()
    for maxd in (1.0, 0.2):
        st = sga.make_setting("ICP", max_correspondence_distance=maxd)
        stats = run_chain(tree, src, st, poses, limit, 1e-6)
        assert stats["warm_passes"] == len(poses) - 1 and stats["walked_points"] > 100, stats


@pytest.mark.parametrize("chunk", [1, 4, 16])
@pytest.mark.parametrize("maxd", [1.0, 0.3, None])
def test_queue_fed_search_equals_lane_search(c1_f32, chunk, maxd):
    """The queue-fed search kernel (a wave refills its lanes from a queue, walks start at the previous neighbour's leaf) and the
    one-query-per-lane kernel return the same canonical neighbours along a pose chain — cold and warm passes, with and without a
    rejector (points without a neighbour remember a leaf instead)."""
    d = c1_f32
    tgt = sga.PointCloud(d["tp"], d["tn"], d["tc"])
    src = sga.PointCloud(d["sp"], d["sn"], d["sc"])
    tree = sga.KdTree(tgt)
    st = sga.make_setting("GICP", max_correspondence_distance=maxd if maxd is not None else 1.0)
    if maxd is None:
        st.factor.max_dist_sq = -1.0
    goal = se3([0.1, 0.2, 1.0], np.deg2rad(0.7), [0.49, 0.12, -0.02])
    pq, pl = sga.Problem(tree, src), sga.Problem(tree, src)
    try:
        for k, T in enumerate(pose_chain(goal) + [se3([1, 0, 0], 0.3, [2.0, 1.0, 0.5]), np.eye(4)]):
            sga.set_search_mode(True, chunk, chunk)
            Hq, bq, eq, nq = pq.linearize(st.factor, T)
            cq, _ = pq.factors()
            sga.set_search_mode(False)
            Hl, bl, el, nl = pl.linearize(st.factor, T)
            cl, _ = pl.factors()
            assert (cq == cl).all(), (k, int((cq != cl).sum()))
            assert nq == nl and np.abs(Hq - Hl).max() <= 1e-6 * np.abs(Hl).max() and abs(eq - el) <= 1e-6 * abs(el), k
        assert pq.pass_stats()["warm_passes"] >= 4
    finally:
        sga.set_search_mode(2, 4, 4)


_ORDER_SCRIPT = r"""
import hashlib, sys
import os

import numpy as np
import small_gicp_amd as sga
target, source, T_gt = sga.synthetic.registration_pair(600_000)
tgt, src = sga.PointCloud(target), sga.PointCloud(source)
sga.estimate_covariances(tgt, None, 10)
sga.estimate_covariances(src, None, 10)
pb = sga.Problem(sga.KdTree(tgt), src)
st = sga.make_setting("GICP", max_correspondence_distance=1.0)
h = hashlib.sha256()
T = np.eye(4)
for step in (0.0, 0.3, 0.05, 0.03, 0.004):  # cold, cold, warm (one query per lane, in the recorded order), warm, warm (queue-fed)
    T = T.copy()
    T[0, 3] += step
    H, b, e, n = pb.linearize(st.factor, T)
    h.update(np.ascontiguousarray(H).tobytes() + np.ascontiguousarray(b).tobytes() + np.float64(e).tobytes() + np.int64(n).tobytes())
print("DIGEST", h.hexdigest(), pb.pass_stats())
"""


@pytest.mark.gpu
def test_launch_order_of_the_tiles_does_not_change_the_results():
    """Longest tile first (linearize.hip: tile_order_kernel): the one-query-per-lane search kernel starts the tiles of a pass in the order
    of the durations recorded by the previous pass.  Partial rows are indexed by tile, so H, b, e and the inlier count must be
    BIT-identical with the feature off (SGA_LPT=0), on for warm passes (1, the default) and on for every pass (2)."""
    import subprocess
    import sys

    digests = []
    for mode in ("0", "1", "2"):
        env = dict(os.environ, SGA_LPT=mode, PYTHONPATH=os.pathsep.join([os.path.dirname(os.path.dirname(os.path.abspath(__file__)))] + sys.path))
        p = subprocess.run([sys.executable, "-c", _ORDER_SCRIPT], capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        digests.append([ln for ln in p.stdout.splitlines() if ln.startswith("DIGEST")][0])
    assert digests[0] == digests[1] == digests[2], digests


def test_exactness_soak_on_random_pose_chains():
    """scripts/soak_exactness.py (short form): random LM-like pose chains on 70k ... 1M-point pairs, GICP and ICP, three rejector
    distances; after every pass the product path's correspondences equal those of a problem that walks every point in every pass."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "scripts", "soak_exactness.py"), "2", "7"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "SOAK OK" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]
    print(p.stdout.splitlines()[-1])


def test_fast_leaf_scan_equals_exact_scan_on_adversarial_inputs(tmp_path):
    """scripts/stress_fast_scan.py: uniform volume, two planes 0.1 mm apart, an integer lattice with duplicates, a cloud 12 km from the
    origin, millimetre scale, dense blobs with far outliers and no rejector, 63 points, one point — the correspondences with the fast leaf
    scan (32-bit truncated keys + exact repeat of undecided queries, the default) equal those with 64-bit keys throughout (SGA_FAST_SCAN=0)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "scripts", "stress_fast_scan.py")
    for mode in ("1", "0"):
        p = subprocess.run([sys.executable, script, str(tmp_path / ("f%s.npz" % mode))], capture_output=True, text=True, timeout=600, env=dict(os.environ, SGA_FAST_SCAN=mode))
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    p = subprocess.run([sys.executable, script, "--compare", str(tmp_path / "f1.npz"), str(tmp_path / "f0.npz")], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout[-3000:]


_HEADROOM_CHAIN = r"""
import hashlib, json
import numpy as np
from scipy.spatial.transform import Rotation
import small_gicp_amd as sga

target, source, T_gt = sga.synthetic.registration_pair(280_000)  # >= 262 144 source points: the streaming warm kernel takes the millimetre passes
tgt, src = sga.PointCloud(target), sga.PointCloud(source)
sga.estimate_covariances(tgt, None, 10)
sga.estimate_covariances(src, None, 10)
pb = sga.Problem(sga.KdTree(tgt), src)
st = sga.make_setting("GICP", max_correspondence_distance=1.0)
rv = Rotation.from_matrix(T_gt[:3, :3]).as_rotvec()
h, sums = hashlib.sha256(), []
for f in (0.0, 0.6, 0.9, 0.98, 0.996, 0.9992, 0.99985, 0.99997, 1.0, 1.0):
    T = np.eye(4)
    T[:3, :3] = Rotation.from_rotvec(rv * f).as_matrix()
    T[:3, 3] = T_gt[:3, 3] * f
    H, b, e, n = pb.linearize(st.factor, T)
    h.update(np.ascontiguousarray(pb.factors()[0]).tobytes())
    sums.append([float(e), int(n), float(np.abs(H).max())])
s = pb.pass_stats()
print("RESULT " + json.dumps({"hash": h.hexdigest(), "sums": sums, "warm": s["warm_passes"], "walked": s["walked_points"]}))
"""


def test_certificate_headroom_changes_nothing_but_the_walkers():
    """SGA_CERT_PAD (linearize.hip: certify) only decides which points of a warm pass search again — every one of them is found again exactly —
    so a pose chain shaped like an LM run gives the same correspondences on every pass whatever the value: the plain check (0), the default,
    and a value that sends nearly every point into the walk.  The switch is read when the library loads, hence one process per value."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for pad in ("0", "0.6", "5"):
        env = dict(os.environ, SGA_CERT_PAD=pad, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        r = subprocess.run([sys.executable, "-c", _HEADROOM_CHAIN], env=env, cwd=root, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (pad, r.stdout[-2000:], r.stderr[-2000:])
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
        out[pad] = json.loads(line[len("RESULT "):])
    print({k: (v["warm"], v["walked"]) for k, v in out.items()})
    ref = out["0"]
    assert ref["warm"] >= 4, ref
    for pad, v in out.items():
        assert v["hash"] == ref["hash"], (pad, "correspondences differ")
        assert v["warm"] == ref["warm"]
        for (e, n, hm), (e0, n0, hm0) in zip(v["sums"], ref["sums"]):
            assert n == n0 and abs(e - e0) <= 1e-6 * abs(e0) and abs(hm - hm0) <= 1e-6 * hm0, (pad, e, e0, n, n0)
    assert out["5"]["walked"] > out["0"]["walked"], out  # (the large value does send more points into the walk)


def test_ring_margin_rule_changes_nothing_but_the_walkers():
    """SGA_GRID_WALK bit 32 (linearize.hip: the margin rule) only decides whether a walker whose ring-1 certificate ends at the ring's face,
    with less than the re-walk's slack beyond its neighbour, keeps that certificate or walks the tree once for a wider one — the neighbour
    is the exact one either way.  Same correspondences and sums on every pass of an LM-shaped pose chain with the rule on (the default)
    and off; with it on, no more points walk in total (the late passes lose their repeat walkers).  One process per value."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for bits in ("15", "47"):
        env = dict(os.environ, SGA_GRID_WALK=bits, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        r = subprocess.run([sys.executable, "-c", _HEADROOM_CHAIN], env=env, cwd=root, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (bits, r.stdout[-2000:], r.stderr[-2000:])
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
        out[bits] = json.loads(line[len("RESULT "):])
    print({k: (v["warm"], v["walked"]) for k, v in out.items()})
    ref = out["15"]
    assert ref["warm"] >= 4, ref
    v = out["47"]
    assert v["hash"] == ref["hash"], "correspondences differ"
    assert v["warm"] == ref["warm"]
    for (e, n, hm), (e0, n0, hm0) in zip(v["sums"], ref["sums"]):
        assert n == n0 and abs(e - e0) <= 1e-6 * abs(e0) and abs(hm - hm0) <= 1e-6 * hm0, (e, e0, n, n0)
    assert v["walked"] <= ref["walked"], out
