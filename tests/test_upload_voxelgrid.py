"""Round 6: the upload and the voxel grid of a preprocessing chain (registration_helper.cpp:22-34; util/downsampling.hpp:23-78) without
copy commands or stream synchronisations (csrc/notes.hpp), with the bounding box taken in the upload's one pass, short sort keys when
the box is known, and pinned host arrays read in place (sga_host_alloc).  Every form must give the records and the voxel partition of
the plain one — bit for bit — and the partition must be the oracle's."""
import ctypes as C

import numpy as np
import pytest

import small_gicp_amd as sga

pytestmark = pytest.mark.gpu


def _scan(n, seed, extent=60.0):
    rng = np.random.default_rng(seed)
    p = rng.normal(0.0, extent / 3, (n, 3)).astype(np.float32)
    p[:, 2] = rng.uniform(-2.0, 6.0, n).astype(np.float32)
    return p


def test_pinned_arrays_are_read_in_place_and_equal_the_staged_upload():
    pts = _scan(50_000, 1)
    nrm = np.random.default_rng(2).normal(size=(len(pts), 3)).astype(np.float32)
    cov = np.random.default_rng(3).uniform(0.1, 1.0, (len(pts), 6)).astype(np.float32)
    a = sga.PointCloud(pts, normals=nrm, covs=cov)
    pp, pn, pc = sga.pinned_copy(pts), sga.pinned_copy(nrm), sga.pinned_copy(cov)
    b = sga.PointCloud(pp, normals=pn, covs=pc)
    c = sga.PointCloud(pp)  # points only
    assert (a.xyz() == pts).all() and (b.xyz() == pts).all() and (c.xyz() == pts).all()
    assert (a.normals()[:, :3] == nrm).all() and (b.normals()[:, :3] == nrm).all()
    assert (a.covs() == b.covs()).all()
    assert (a.origin() == 0).all() and (b.origin() == 0).all()
    # a partly pinned set of arrays takes the staged path
    d = sga.PointCloud(pp, normals=nrm)
    assert (d.xyz() == pts).all() and (d.normals()[:, :3] == nrm).all()


def test_origin_is_chosen_from_the_box_of_either_upload_path():
    """device frames: a cloud far from the origin is recentred in double about the centre of its bounding box rounded to 128 m —
    the box comes from the staging pass (pageable) or from the pack kernel as a note (pinned); non-finite coordinates are ignored."""
    pts = _scan(30_000, 4) + np.array([70_000.0, -3_000.0, 200.0], np.float32)
    pts[17] = [np.nan, np.inf, -np.inf]
    a, b = sga.PointCloud(pts), sga.PointCloud(sga.pinned_copy(pts))
    fin = np.isfinite(pts).all(axis=1)
    centre = 0.5 * (pts[fin].astype(np.float64).min(axis=0) + pts[fin].astype(np.float64).max(axis=0))
    for cl in (a, b):
        assert (cl.origin() == 128.0 * np.round(centre / 128.0)).all(), (cl.origin(), centre)
    xa, xb = a.xyz64(), b.xyz64()
    assert np.array_equal(xa, xb, equal_nan=True)
    assert np.abs(xa[fin] - pts[fin].astype(np.float64)).max() < 1e-6  # fp32 records relative to the origin hold the fp32 inputs exactly here


@pytest.mark.parametrize("leaf", [0.1, 0.25, 1.0])
def test_voxelgrid_short_keys_equal_reference_keys_and_the_oracle(orc, leaf):
    """An uploaded cloud (box known: short keys, 32-bit sort) against the same points as a cloud made on the device (a slice: no box, the
    reference's 63-bit keys) and against the oracle: same voxels in the same order, same centroids."""
    pts = _scan(120_000, 5)
    up = sga.PointCloud(pts)
    dev = up.slice(0, len(pts))
    a, b = sga.voxelgrid_sampling(up, leaf).xyz(), sga.voxelgrid_sampling(dev, leaf).xyz()
    ref = orc.voxelgrid_sampling(pts, leaf)
    assert a.shape == b.shape == ref.shape
    assert (a == b).all()
    assert np.abs(a - ref).max() < 1e-5


def test_voxelgrid_drops_what_the_reference_drops(orc):
    """downsampling.hpp:41-46: points outside +-2^20 voxels are dropped (with a warning there); non-finite ones never had a voxel."""
    pts = _scan(20_000, 6)
    pts[5] = [3e5, 0, 0]      # 3e5 / 0.25 = 1.2e6 > 2^20: dropped
    pts[6] = [0, -2.7e5, 1]   # dropped
    pts[7] = [np.nan, 0, 0]
    pts[8] = [-3e5, 0, 0]     # (the mirror images keep the centre of the box, and with it the device frame's origin, at 0)
    pts[9] = [0, 2.7e5, 1]
    keep = np.ones(len(pts), bool)
    keep[[5, 6, 7, 8, 9]] = False
    for cloud in (sga.PointCloud(pts), sga.PointCloud(pts).slice(0, len(pts))):
        out = sga.voxelgrid_sampling(cloud, 0.25).xyz()
        ref = orc.voxelgrid_sampling(pts[keep], 0.25)
        assert out.shape == ref.shape and np.abs(out - ref).max() < 1e-4
    # every point dropped: an empty cloud, not an error
    far = np.full((100, 3), 1e9, np.float32)
    assert sga.voxelgrid_sampling(sga.PointCloud(far), 0.25).size() == 0


def test_voxelgrid_large_cloud_and_many_tiles(orc):
    """More than 64 tiles of 2048 keys: the look-back of ds_segments_kernel crosses its 64-tile window; more than 262144 points: the
    centroid kernel waits for the count instead of being launched for the largest possible one."""
    pts = _scan(400_000, 7, extent=150.0)
    for leaf in (0.5, 0.05):
        out = sga.voxelgrid_sampling(pts, leaf).xyz()
        ref = orc.voxelgrid_sampling(pts, leaf)
        assert out.shape == ref.shape and np.abs(out - ref).max() < 1e-4
    # one voxel for everything / one voxel per point
    one = sga.voxelgrid_sampling(pts[:5000] * 1e-3, 10.0).xyz()
    assert len(one) <= 8
    allv = sga.voxelgrid_sampling(pts[:70_000], 1e-3)  # (|p| / leaf stays below 2^20: nothing is dropped)
    assert allv.size() == len(np.unique(np.floor(pts[:70_000].astype(np.float64) * (1.0 / 1e-3)).astype(np.int64), axis=0))


def test_stream_ordered_chain_gives_the_same_cloud():
    """sga_context_set_stream_ordered: the upload returns with its kernel in flight (the staging slot is recycled behind an event), the
    voxel grid waits for nothing but its own count — the chain's result is the synchronous one, for many scans in a row (ring reuse)."""
    ref_ctx, ctx = sga.Context(0), sga.Context(0)
    ctx.set_stream_ordered(True)
    for f in range(8):
        pts = _scan(60_000 + 1000 * f, 10 + f)
        a = sga.voxelgrid_sampling(sga.PointCloud(pts, ctx=ref_ctx), 0.25)
        b = sga.voxelgrid_sampling(sga.PointCloud(pts, ctx=ctx), 0.25)
        c = sga.voxelgrid_sampling(sga.PointCloud(sga.pinned_copy(pts), ctx=ctx), 0.25)
        xa = a.xyz()
        assert a.size() == b.size() == c.size() and (xa == b.xyz()).all() and (xa == c.xyz()).all()
    ctx.set_stream_ordered(False)


def test_host_alloc_roundtrip():
    lib = sga.load()
    p = C.c_void_p()
    assert lib.sga_host_alloc(1 << 20, C.byref(p)) == 0 and p.value
    C.memset(p, 7, 1 << 20)
    assert lib.sga_host_free(p) == 0
    q = C.c_void_p(123)
    assert lib.sga_host_alloc(0, C.byref(q)) == 0 and not q.value
    assert lib.sga_host_free(None) == 0
