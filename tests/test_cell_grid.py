"""The cell-grid search (csrc/cell_grid.hpp / cell_grid.hip) returns the SAME correspondences as the kd walk — the exact nearest neighbour of
KdTree::nearest_neighbor_search (ann/kdtree.hpp:193-233, knn_result.hpp:80-100) under the canonical tie rule — and writes certificates
(nn / nn2 / rex) the warm passes that follow can rely on.  On a real MI355X through the C-ABI.

Both searches feed the same factor stage, so equal correspondences give equal sums up to the summation order of the two factor kernels
(fused into the kd search / separate after the grid search): 1e-6 relative in fp32 arithmetic.
"""
import numpy as np
import pytest

import small_gicp_amd as sga
from test_warm_pass import pose_chain, se3

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def restore_modes():
    lim = sga.get_warm_limit()
    yield
    sga.set_warm_limit(lim)
    sga.set_grid_mode(1, 65536)


def both(tree, src, setting, poses, rel=2e-6, grid_mode=4):
    """linearize along `poses` on two problems — one searched by the grid, one by the kd walk only — and compare everything."""
    pg, pk = sga.Problem(tree, src), sga.Problem(tree, src)
    for k, T in enumerate(poses):
        sga.set_grid_mode(grid_mode)
        Hg, bg, eg, ng = pg.linearize(setting.factor, T)
        e2g = pg.error(setting.factor, T)
        cg, _ = pg.factors()
        sga.set_grid_mode(0)
        Hk, bk, ek, nk = pk.linearize(setting.factor, T)
        ck, _ = pk.factors()
        assert (cg == ck).all(), (k, int((cg != ck).sum()), np.nonzero(cg != ck)[0][:8], cg[cg != ck][:8], ck[cg != ck][:8])
        assert ng == nk, (k, ng, nk)
        scale = max(np.abs(Hk).max(), 1e-30)
        assert np.abs(Hg - Hk).max() <= rel * scale and np.abs(bg - bk).max() <= rel * max(np.abs(bk).max(), scale * 1e-3), k
        assert abs(eg - ek) <= rel * max(abs(ek), 1e-30) and abs(e2g - eg) <= 1e-5 * max(abs(eg), 1e-30), k
    sg, sk = pg.pass_stats(), pk.pass_stats()
    assert sk["grid_passes"] == 0
    return sg


@pytest.mark.parametrize("kind", ["GICP", "PLANE_ICP", "ICP"])
@pytest.mark.parametrize("mode", ["fp32", "fp64"])
def test_grid_equals_kd_on_c1(c1_f32, kind, mode):
    """C1 (real scans, 6k points after the voxel grid) with a grid forced onto the small target: every pass of an LM-shaped pose chain."""
    d = c1_f32
    sga.set_grid_mode(4, 16)
    tgt = sga.PointCloud(d["tp"], d["tn"], d["tc"])
    src = sga.PointCloud(d["sp"], d["sn"], d["sc"])
    tree = sga.KdTree(tgt)
    st = sga.make_setting(kind, math_mode=mode)
    goal = se3([0.1, 0.2, 1.0], np.deg2rad(0.7), [0.49, 0.12, -0.02])
    poses = pose_chain(goal)
    stats = both(tree, src, st, poses, 2e-6 if mode == "fp32" else 1e-12)
    assert stats["grid_passes"] == len(poses) and stats["grid_cell_m"] > 0, stats


@pytest.mark.parametrize("max_dist", [0.05, 0.3, 1.0, 2.0])
def test_grid_equals_kd_on_the_synthetic_scene(max_dist):
    """200k <-> 200k points of the benchmark scene (walls six times denser than the ground, 10 % clutter far from everything): far pose,
    near pose, optimum; rejector reach from a fifth of a cell to ten cells."""
    target, source, T_gt = sga.synthetic.registration_pair(200_000)
    sga.set_grid_mode(4, 16)
    tgt, src = sga.PointCloud(target), sga.PointCloud(source)
    tree = sga.KdTree(tgt)
    st = sga.make_setting("ICP", max_correspondence_distance=max_dist)
    near = T_gt @ se3([0.3, 0.5, 0.8], np.deg2rad(0.15), [0.03, -0.03, 0.02])
    stats = both(tree, src, st, [np.eye(4), near, T_gt, T_gt])
    assert stats["grid_passes"] == 4, stats


def test_grid_certificates_serve_the_warm_passes():
    """mode 3: the first pass and the cold passes go through the grid, the passes after small motions are warm and consume the
    certificates the grid wrote — compared with cold kd walks at every pose."""
    target, source, T_gt = sga.synthetic.registration_pair(150_000)
    sga.set_grid_mode(3, 16)
    tgt, src = sga.PointCloud(target), sga.PointCloud(source)
    sga.estimate_covariances(tgt, None, 10)
    sga.estimate_covariances(src, None, 10)
    tree = sga.KdTree(tgt)
    st = sga.make_setting("GICP", max_correspondence_distance=1.0)
    poses = pose_chain(T_gt, fractions=(0.0, 0.9, 0.99, 0.998, 0.9995, 0.9999, 1.0, 1.0))
    pg, pk = sga.Problem(tree, src), sga.Problem(tree, src)
    for k, T in enumerate(poses):
        sga.set_grid_mode(3)
        sga.set_warm_limit(0.1)
        Hg, bg, eg, ng = pg.linearize(st.factor, T)
        cg, _ = pg.factors()
        sga.set_grid_mode(0)
        sga.set_warm_limit(-1.0)
        Hk, bk, ek, nk = pk.linearize(st.factor, T)
        ck, _ = pk.factors()
        assert (cg == ck).all(), (k, int((cg != ck).sum()))
        assert ng == nk and abs(eg - ek) <= 2e-6 * abs(ek), (k, ng, nk, eg, ek)
    s = pg.pass_stats()
    assert s["grid_passes"] >= 2 and s["warm_passes"] >= 3, s
    assert s["walked_points"] < 0.2 * len(source) * s["warm_passes"], s  # the grid's certificates held for most points


def test_grid_ties_follow_the_canonical_rule():
    """A lattice target and queries at equal distance from 2, 4 or 8 lattice points: the neighbour is the one of lowest kd position for
    both searches (knn_result.hpp:81-83 leaves ties to the traversal order; ours is canonical, kd_search.hpp)."""
    g = np.arange(0, 40, dtype=np.float32) * 0.25
    X, Y, Z = np.meshgrid(g, g, g[:12], indexing="ij")
    target = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1).astype(np.float32)
    rng = np.random.default_rng(5)
    base = target[rng.integers(0, len(target), 60_000)]
    off = rng.choice([0.0, 0.125], size=base.shape).astype(np.float32)  # exactly between lattice points along a random subset of the axes
    source = (base + off).astype(np.float32)
    sga.set_grid_mode(4, 16)
    tgt, src = sga.PointCloud(target), sga.PointCloud(source)
    tree = sga.KdTree(tgt)
    st = sga.make_setting("ICP", max_correspondence_distance=1.0)
    both(tree, src, st, [np.eye(4), se3([0, 0, 1], 0.0, [0.25, 0.0, 0.0]), se3([0, 0, 1], 0.0, [0.125, 0.125, 0.0])])


def test_grid_queries_outside_the_target_box_and_without_neighbours():
    """A source three times the extent of the target (most queries lie outside the grid, many beyond the rejector's reach)."""
    rng = np.random.default_rng(11)
    target = (rng.uniform(-5, 5, size=(120_000, 3)) * np.array([1.0, 1.0, 0.02])).astype(np.float32)  # a 10 m x 10 m slab
    source = (rng.uniform(-15, 15, size=(80_000, 3)) * np.array([1.0, 1.0, 0.1])).astype(np.float32)
    sga.set_grid_mode(4, 16)
    tgt, src = sga.PointCloud(target), sga.PointCloud(source)
    tree = sga.KdTree(tgt)
    for md in (0.2, 1.5):
        st = sga.make_setting("ICP", max_correspondence_distance=md)
        both(tree, src, st, [np.eye(4), se3([0.2, 0.1, 1], np.deg2rad(20), [1.0, -2.0, 0.3])])


def test_registration_result_does_not_depend_on_the_search():
    """A whole C3-shaped registration (300k points, GICP, mode 2: the grid from the second pass on) against kd walks only."""
    target, source, T_gt = sga.synthetic.registration_pair(300_000)
    sga.set_grid_mode(2)  # (the index gets its grid when it is built)
    tgt, src = sga.PointCloud(target), sga.PointCloud(source)
    sga.estimate_covariances(tgt, None, 20)
    sga.estimate_covariances(src, None, 20)
    tree = sga.KdTree(tgt)
    st = sga.make_setting("GICP", max_correspondence_distance=1.0)
    pg = sga.Problem(tree, src)
    rg = pg.align(st, np.eye(4))
    sga.set_grid_mode(0)
    pk = sga.Problem(tree, src)
    rk = pk.align(st, np.eye(4))
    assert pg.pass_stats()["grid_passes"] >= 1 and pk.pass_stats()["grid_passes"] == 0
    assert rg.iterations == rk.iterations and rg.num_inliers == rk.num_inliers
    assert np.abs(rg.T_target_source - rk.T_target_source).max() < 1e-7
    assert abs(rg.error - rk.error) <= 1e-6 * abs(rk.error)
