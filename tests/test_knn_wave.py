"""Round 6: normals / covariances of clouds that do not fill the chip are estimated with ONE WAVE PER QUERY (csrc/knn_wave.hpp) instead
of one query per lane (kd_search.hpp: kd_knn_own_points / kd_knn).  Both searches are exact (util/normal_estimation.hpp:65-92 over
ann/kdtree.hpp:172-176, the point itself included): on clouds without equidistant neighbours they must select the same sets, so the
results agree to the rounding of the fp64 sums (the order of the additions differs: nearest first here, as in the reference)."""
import numpy as np
import pytest

import small_gicp_amd as sga
from small_gicp_amd import synthetic

pytestmark = pytest.mark.gpu


def _features(pts, k, wave):
    lib = sga.load()
    lib.sga_set_knn_wave_max(1 << 40 if wave else 0)
    try:
        c = sga.PointCloud(pts)
        sga.estimate_normals_covariances(c, None, k)
        return c.normals()[:, :3], c.covs()[:, :3, :3]
    finally:
        lib.sga_set_knn_wave_max(81920)


def _brute_features(pts, k):
    p = pts.astype(np.float64)
    d2 = ((p[:, None, :] - p[None, :, :]) ** 2).sum(-1)
    idx = np.argsort(d2, axis=1, kind="stable")[:, :k]
    nb = p[idx]
    mean = nb.mean(axis=1)
    cov = np.einsum("nki,nkj->nij", nb, nb) / nb.shape[1] - np.einsum("ni,nj->nij", mean, mean)
    w, v = np.linalg.eigh(cov)
    n0 = v[:, :, 0]
    flip = (p * n0).sum(-1) > 0
    n0[flip] *= -1
    return n0, w


@pytest.mark.parametrize("k", [5, 10, 20, 33, 64])
def test_wave_search_equals_lane_search(k, c1_raw):
    scan = synthetic.kitti_like_scan(3)[0]
    down = sga.voxelgrid_sampling(scan, 0.25).xyz()
    for pts in (down, sga.voxelgrid_sampling(c1_raw[0], 0.25).xyz()):
        nw, cw = _features(pts, k, True)
        nl, cl = _features(pts, k, False)
        dn = np.abs(nw - nl).max(axis=1)
        dc = np.abs(cw - cl).reshape(len(pts), -1).max(axis=1)
        bad = (dn > 1e-3) | (dc > 1e-3)  # a different neighbour set (equidistant k-th candidates) or a near-degenerate spectrum
        print("k=%d n=%d: %d points differ between the two searches, the rest to %.1e / %.1e" % (k, len(pts), bad.sum(), dn[~bad].max(), dc[~bad].max()))
        assert bad.sum() <= 2 and dn[~bad].max() < 1e-5 and dc[~bad].max() < 1e-5


@pytest.mark.parametrize("n,k", [(33000, 20), (70000, 10), (81920, 20)])
def test_wave_search_equals_lane_search_at_the_upper_sizes(n, k):
    """The sizes the one-wave-per-query search took over late in round 6 (32 768 < n <= 81 920): the same neighbour sets as the
    one-query-per-lane search on the synthetic scene."""
    pts = synthetic.scene(n, 4)[:, :3].astype(np.float32)
    nw, cw = _features(pts, k, True)
    nl, cl = _features(pts, k, False)
    dn = np.abs(nw - nl).max(axis=1)
    dc = np.abs(cw - cl).reshape(len(pts), -1).max(axis=1)
    bad = (dn > 1e-3) | (dc > 1e-3)
    print("n=%d k=%d: %d points differ between the two searches, the rest to %.1e / %.1e" % (n, k, bad.sum(), dn[~bad].max(), dc[~bad].max()))
    assert bad.sum() <= max(2, n // 20000) and dn[~bad].max() < 1e-5 and dc[~bad].max() < 1e-5


@pytest.mark.parametrize("n", [1, 4, 5, 8, 9, 63, 64, 65, 127, 500, 513, 4097])
def test_small_and_odd_sizes_against_brute_force(n):
    """Fewer points than k, fewer than a batch (the tree is one batch), sizes around the batch and leaf boundaries: the wave search against
    a brute-force kNN (normal up to sign convention, eigenvalues of the neighbourhood covariance through C = V diag(1e-3, 1, 1) V^T)."""
    rng = np.random.default_rng(n)
    pts = (rng.normal(size=(n, 3)) * [3.0, 2.0, 0.3] + [10.0, -4.0, 1.0]).astype(np.float32)
    k = 10
    nw, cw = _features(pts, k, True)
    nl, cl = _features(pts, k, False)
    assert np.abs(nw - nl).max() < 1e-5 and np.abs(cw - cl).max() < 1e-5
    if n >= 5:  # normal_estimation.hpp:33-37: fewer than 5 neighbours -> zero normal, identity covariance
        nb, _ = _brute_features(pts, min(k, n))
        cosang = np.abs((nw * nb).sum(-1))
        assert (cosang > 1 - 1e-4).sum() >= n - max(1, n // 100), cosang.min()  # (near-isotropic neighbourhoods: the smallest eigenvector is ill-conditioned)
    else:
        assert not nw.any() and np.abs(cw - np.eye(3)).max() == 0


def test_non_finite_points_are_nobodys_neighbour():
    rng = np.random.default_rng(5)
    pts = rng.uniform(-5, 5, (3000, 3)).astype(np.float32)
    clean_n, clean_c = _features(pts, 10, True)
    # (a kd-tree over non-finite coordinates is refused by the index build: the box reports them — same as before this round)
    bad = pts.copy()
    bad[7] = [np.nan, 0, 0]
    with pytest.raises(sga.SgaError):
        _features(bad, 10, True)
    assert np.isfinite(clean_n).all() and np.isfinite(clean_c).all()
