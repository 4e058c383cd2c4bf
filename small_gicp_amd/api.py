"""Python host layer over the C-ABI, mirroring the reference's Python module surface
(src/python/{pointcloud,kdtree,voxelmap,preprocess,align,result}.cpp of /root/reference): PointCloud, KdTree,
GaussianVoxelMap, voxelgrid_sampling, estimate_*, preprocess_points, align, RegistrationResult.
All compute happens in the HIP library; numpy only carries host buffers in and out.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import GICP, ICP, PLANE_ICP, FactorParams, RegistrationSettingC, ResultC, check, load

_FACTOR_BY_NAME = {"ICP": ICP, "PLANE_ICP": PLANE_ICP, "GICP": GICP, "VGICP": GICP}


def _fp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _T16(T):
    T = np.eye(4) if T is None else np.asarray(T, dtype=np.float64).reshape(4, 4)
    return np.ascontiguousarray(T.T).reshape(16)  # column-major


class Context:
    """One GPU + one HIP stream (sga_context)."""

    def __init__(self, device=0, stream=None):
        self.h = C.c_void_p()
        L = load()
        if stream is None:
            check(L.sga_context_create(int(device), C.byref(self.h)))
        else:
            check(L.sga_context_create_on_stream(int(device), C.c_void_p(int(stream)), C.byref(self.h)))
        self.device = device

    def __del__(self):
        if getattr(self, "h", None) and self.h.value:
            load().sga_context_destroy(self.h)
            self.h = C.c_void_p()

    def synchronize(self):
        check(load().sga_context_synchronize(self.h))

    @staticmethod
    def comm_unique_id():
        """128-byte RCCL unique id (create on rank 0, hand to the other ranks out of band)."""
        buf = (C.c_ubyte * 128)()
        check(load().sga_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, nranks, rank, unique_id):
        """Join the RCCL communicator: afterwards linearize/error/align all-reduce their accumulators over the ranks."""
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        check(load().sga_comm_init(self.h, int(nranks), int(rank), buf))

    def comm_init_callback(self, nranks, rank, allreduce):
        """The same protocol with the caller's transport: allreduce(values) must return the element-wise sum over all ranks of the
        numpy array it is given (e.g. a torch.distributed / MPI all-reduce on the host)."""
        from ._lib import ALLREDUCE_FN

        def _cb(_user, ptr, count):
            try:
                a = np.ctypeslib.as_array(ptr, shape=(count,))
                a[:] = np.asarray(allreduce(a.copy()), dtype=np.float64).reshape(count)
                return 0
            except Exception:  # noqa: BLE001
                import traceback

                traceback.print_exc()
                return 1

        self._allreduce_cb = ALLREDUCE_FN(_cb)  # keep alive as long as the context
        check(load().sga_comm_init_callback(self.h, int(nranks), int(rank), self._allreduce_cb, None))

    def comm_destroy(self):
        check(load().sga_comm_destroy(self.h))

    def set_stream_ordered(self, enabled=True):
        """Index builds and normal / covariance estimation return once enqueued (results ordered for later calls on this context; a
        consumer on another context waits for the producer's event).  Returns the previous mode."""
        prev = getattr(self, "stream_ordered", False)
        check(load().sga_context_set_stream_ordered(self.h, int(enabled)))
        self.stream_ordered = bool(enabled)
        return prev

    def gpu_time_ms(self, fn):
        """GPU time (HIP events on the context's stream) of whatever fn() enqueues; returns (milliseconds, fn's result)."""
        check(load().sga_debug_timer_start(self.h))
        r = fn()
        ms = C.c_double()
        check(load().sga_debug_timer_stop(self.h, C.byref(ms)))
        return ms.value, r

    def set_profiling(self, enabled=True):
        check(load().sga_context_set_profiling(self.h, int(enabled)))

    def kernel_ms(self):
        lm, em = C.c_double(), C.c_double()
        lc, ec = C.c_uint64(), C.c_uint64()
        check(load().sga_context_get_kernel_ms(self.h, C.byref(lm), C.byref(lc), C.byref(em), C.byref(ec)))
        sm, sc = C.c_double(), C.c_uint64()
        check(load().sga_context_get_search_ms(self.h, C.byref(sm), C.byref(sc)))
        cm, wm = C.c_double(), C.c_double()
        cc, wc = C.c_uint64(), C.c_uint64()
        wf = C.c_double()
        check(load().sga_context_get_pass_ms(self.h, C.byref(cm), C.byref(cc), C.byref(wm), C.byref(wc), C.byref(wf)))
        km, kc = C.c_double(), C.c_uint64()
        check(load().sga_context_get_comm_ms(self.h, C.byref(km), C.byref(kc)))
        return {"linearize_ms": lm.value, "linearize_calls": lc.value, "error_ms": em.value, "error_calls": ec.value, "search_ms": sm.value, "search_calls": sc.value,
                "cold_ms": cm.value, "cold_calls": cc.value, "warm_ms": wm.value, "warm_calls": wc.value, "warm_search_ms": wf.value,
                "comm_ms": km.value, "comm_calls": kc.value}


_DEFAULT_CTX = None


def default_context():
    global _DEFAULT_CTX
    if _DEFAULT_CTX is None:
        _DEFAULT_CTX = Context(0)
    return _DEFAULT_CTX


def sym6_from_mats(covs):
    """(N,3,3) or (N,4,4) symmetric -> (N,6) xx,xy,xz,yy,yz,zz"""
    c = np.asarray(covs)
    return np.stack([c[:, 0, 0], c[:, 0, 1], c[:, 0, 2], c[:, 1, 1], c[:, 1, 2], c[:, 2, 2]], axis=1)


def mats_from_sym6(c6):
    c6 = np.asarray(c6)
    m = np.empty((len(c6), 3, 3), dtype=c6.dtype)
    m[:, 0, 0], m[:, 0, 1], m[:, 0, 2] = c6[:, 0], c6[:, 1], c6[:, 2]
    m[:, 1, 0], m[:, 1, 1], m[:, 1, 2] = c6[:, 1], c6[:, 3], c6[:, 4]
    m[:, 2, 0], m[:, 2, 1], m[:, 2, 2] = c6[:, 2], c6[:, 4], c6[:, 5]
    return m


class _PinnedBlock:
    """Owner of one sga_host_alloc block (freed when the last array viewing it is gone)."""

    def __init__(self, nbytes):
        self.p = C.c_void_p()
        check(load().sga_host_alloc(int(nbytes), C.byref(self.p)))
        self.nbytes = int(nbytes)

    def __del__(self):
        if getattr(self, "p", None) and self.p.value:
            load().sga_host_free(self.p)
            self.p = C.c_void_p()


def pinned_empty(shape, dtype=np.float32):
    """A numpy array in pinned host memory (sga_host_alloc).  Scans read into such arrays are uploaded without a CPU copy: the device reads
    them in place (PointCloud(points) recognises them) — what a driver that preloads its scans into host memory should use
    (benchmark/benchmark_odom.hpp:36-47 keeps every scan in host memory before the timed loop).  The block is freed with the last array
    that views it."""
    dt = np.dtype(dtype)
    count = int(np.prod(shape)) if np.ndim(shape) else int(shape)
    block = _PinnedBlock(max(1, count * dt.itemsize))
    buf = (C.c_char * block.nbytes).from_address(block.p.value)
    buf._sga_block = block  # np.frombuffer keeps `buf` alive, `buf` keeps the block
    return np.frombuffer(buf, dtype=dt, count=count).reshape(shape)


def pinned_copy(a, dtype=np.float32):
    """`a` copied into pinned host memory (C-contiguous, `dtype`)."""
    a = np.asarray(a)
    out = pinned_empty(a.shape, dtype)
    out[...] = a
    return out


class PointCloud:
    """Device-resident point cloud (points [+ normals] [+ covariances]); mirrors small_gicp.PointCloud."""

    def __init__(self, points=None, normals=None, covs=None, ctx=None, _handle=None):
        self.ctx = ctx or default_context()
        if _handle is not None:
            self.h = _handle
            return
        pts = np.zeros((0, 3), np.float32) if points is None else np.asarray(points)
        if pts.ndim != 2 or pts.shape[1] not in (3, 4):
            raise ValueError("points must be (N,3) or (N,4)")
        nrm = None if normals is None else np.ascontiguousarray(np.asarray(normals)[:, :3], dtype=np.float32)
        c6 = None
        if covs is not None:
            covs = np.asarray(covs)
            c6 = covs if covs.ndim == 2 and covs.shape[1] == 6 else sym6_from_mats(covs)
            c6 = np.ascontiguousarray(c6, dtype=np.float32)
        self.h = C.c_void_p()
        if pts.dtype == np.float64 and len(pts) > 0:
            # double input (the reference's PointCloud is double, points/point_cloud.hpp:69-71): the device keeps fp32 records RELATIVE to an
            # origin (small_gicp_amd.h, "device frames"); the subtraction happens here, in double, so a geo-referenced cloud keeps its millimetres
            p3 = pts[:, :3]
            fin = np.isfinite(p3)
            lo = np.where(fin, p3, np.inf).min(axis=0).astype(np.float64)
            hi = np.where(fin, p3, -np.inf).max(axis=0).astype(np.float64)
            origin = np.zeros(3)
            load().sga_choose_origin(_dp(np.ascontiguousarray(lo)), _dp(np.ascontiguousarray(hi)), _dp(origin))
            rel = np.ascontiguousarray(p3 - origin if origin.any() else p3, dtype=np.float32)
            check(load().sga_cloud_create_f32_origin(self.ctx.h, _fp(rel), _fp(nrm), _fp(c6), len(rel), _dp(origin), C.byref(self.h)))
        else:
            # (an (N,3) float32 C-contiguous array — e.g. one from pinned_empty — goes down as it is: no copy on this side)
            xyz = pts if (pts.shape[1] == 3 and pts.dtype == np.float32 and pts.flags.c_contiguous) else np.ascontiguousarray(pts[:, :3], dtype=np.float32)
            check(load().sga_cloud_create_f32(self.ctx.h, _fp(xyz), _fp(nrm), _fp(c6), len(xyz), C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None) and self.h.value:
            load().sga_cloud_destroy(self.h)
            self.h = C.c_void_p()

    def origin(self):
        """Origin of the cloud's device frame (the device holds fl32(p - origin)); zero for clouds centred within 64 m of the origin."""
        o = np.zeros(3)
        check(load().sga_cloud_origin(self.h, _dp(o)))
        return o

    def size(self):
        n = C.c_size_t()
        check(load().sga_cloud_size(self.h, C.byref(n)))
        return n.value

    __len__ = size

    def point(self, i):
        """pointcloud.cpp: the i-th point as a homogeneous 4-vector (points/point_cloud.hpp:49)."""
        return self.points()[int(i)]

    def normal(self, i):
        return self.normals()[int(i)]

    def cov(self, i):
        return self.covs()[int(i)]

    def slice(self, first, count):
        """A new device cloud holding points [first, first + count) with their normals / covariances (sga_cloud_slice): the source
        shard of one rank when a registration is spread over GPUs."""
        h = C.c_void_p()
        check(load().sga_cloud_slice(self.ctx.h, self.h, int(first), int(count), C.byref(h)))
        return PointCloud(ctx=self.ctx, _handle=h)

    def empty(self):
        return self.size() == 0

    def _has(self):
        a, b = C.c_int(), C.c_int()
        check(load().sga_cloud_has(self.h, C.byref(a), C.byref(b)))
        return bool(a.value), bool(b.value)

    def points(self):
        """(N,4) float64 homogeneous points, like the reference binding."""
        n = self.size()
        xyz = np.empty((n, 3), np.float64)
        check(load().sga_cloud_download_f64(self.ctx.h, self.h, _dp(xyz), None, None))  # device record + origin, added in double
        return np.concatenate([xyz, np.ones((n, 1))], axis=1)

    def xyz64(self):
        n = self.size()
        xyz = np.empty((n, 3), np.float64)
        check(load().sga_cloud_download_f64(self.ctx.h, self.h, _dp(xyz), None, None))
        return xyz

    def xyz(self):
        n = self.size()
        xyz = np.empty((n, 3), np.float32)
        check(load().sga_cloud_download(self.ctx.h, self.h, _fp(xyz), None, None))
        return xyz

    def normals(self):
        n = self.size()
        if not self._has()[0]:
            return np.zeros((n, 4))
        nr = np.empty((n, 3), np.float32)
        check(load().sga_cloud_download(self.ctx.h, self.h, None, _fp(nr), None))
        return np.concatenate([nr.astype(np.float64), np.zeros((n, 1))], axis=1)

    def covs(self):
        """(N,4,4) float64 with zero padding, like the reference binding."""
        n = self.size()
        out = np.zeros((n, 4, 4))
        if self._has()[1]:
            c6 = np.empty((n, 6), np.float32)
            check(load().sga_cloud_download(self.ctx.h, self.h, None, None, _fp(c6)))
            out[:, :3, :3] = mats_from_sym6(c6.astype(np.float64))
        return out


class KdTree:
    """Exact nearest-neighbour index over a PointCloud (small_gicp.KdTree): an implicit balanced kd-tree built on the GPU
    (sga_index_build_kdtree)."""

    def __init__(self, points, num_threads=1, **_ignored):
        if not isinstance(points, PointCloud):
            points = PointCloud(points)
        self.cloud = points
        self.ctx = points.ctx
        self.h = C.c_void_p()
        check(load().sga_index_build_kdtree(self.ctx.h, points.h, C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None) and self.h.value:
            load().sga_index_destroy(self.h)
            self.h = C.c_void_p()

    def size(self):
        n = C.c_size_t()
        check(load().sga_index_size(self.h, C.byref(n)))
        return n.value

    __len__ = size

    def spacing(self):
        """The index's own length scale (geometric mean leaf diagonal; sga_index_spacing): what the pass routing measures motions in."""
        v = C.c_double()
        check(load().sga_index_spacing(self.h, C.byref(v)))
        return v.value

    def refresh_attributes(self):
        """Pull the cloud's current normals / covariances into the index (needed when they were set after the index was built)."""
        check(load().sga_index_refresh_attributes(self.ctx.h, self.h, self.cloud.h))

    def batch_knn_search(self, pts, k, max_sq_dist=-1.0, num_threads=1):
        """kdtree.cpp:128-205: (indices (m,k) int64, squared distances (m,k) float64) — double distances like the reference's
        (sga_index_knn_f64: the search runs in fp32, the distances of the neighbours found are evaluated in double)."""
        q = np.ascontiguousarray(np.asarray(pts)[:, :3], dtype=np.float64)
        idx = np.empty((len(q), k), np.int64)
        d2 = np.empty((len(q), k), np.float64)
        check(load().sga_index_knn_f64(self.ctx.h, self.h, _dp(q), len(q), int(k), float(max_sq_dist), idx.ctypes.data_as(C.POINTER(C.c_int64)), _dp(d2)))
        return idx, d2

    def batch_nearest_neighbor_search(self, pts, num_threads=1):
        idx, d2 = self.batch_knn_search(pts, 1)
        return idx[:, 0], d2[:, 0]

    def knn_search(self, pt, k):
        idx, d2 = self.batch_knn_search(np.asarray(pt, dtype=np.float64).reshape(1, -1), k)
        return idx[0], d2[0]

    def nearest_neighbor_search(self, pt):
        idx, d2 = self.knn_search(pt, 1)
        return (1 if idx[0] >= 0 else 0), int(idx[0]), float(d2[0])


class GaussianVoxelMap:
    """small_gicp.GaussianVoxelMap (src/python/voxelmap.cpp:20-140): `GaussianVoxelMap(leaf_size)`, then any number of
    `insert(cloud_with_covs, T)`; `set_lru(horizon, clear_cycle)`; `size()`, `voxel_points()`, `voxel_covs()`.
    Device side: the incremental map of csrc/voxelmap.hip."""

    def __init__(self, leaf_size, ctx=None):
        self.leaf = float(leaf_size)
        self.ctx = ctx or default_context()
        self.h = C.c_void_p()
        check(load().sga_voxelmap_create(self.ctx.h, self.leaf, C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None) and self.h.value:
            load().sga_index_destroy(self.h)
            self.h = C.c_void_p()

    @classmethod
    def from_voxels(cls, leaf_size, coords, means, cov6, ctx=None):
        """A map from voxels that exist on the host in the reference's flat order (sga_index_create_voxelmap_from_voxels: what
        ParallelReductionHIP uploads for a GaussianVoxelMap target).  A search target only: insert() needs the running sums."""
        self = cls.__new__(cls)
        self.leaf = float(leaf_size)
        self.ctx = ctx or default_context()
        self.h = C.c_void_p()
        coords = np.ascontiguousarray(coords, dtype=np.int32).reshape(-1, 3)
        means = np.ascontiguousarray(means, dtype=np.float64).reshape(-1, 3)
        cov6 = np.ascontiguousarray(cov6, dtype=np.float64).reshape(-1, 6)
        if not (len(coords) == len(means) == len(cov6)):
            raise ValueError("coords, means and cov6 must have one row per voxel")
        check(load().sga_index_create_voxelmap_from_voxels(self.ctx.h, self.leaf, coords.ctypes.data_as(C.c_void_p), _dp(means), _dp(cov6), len(coords), C.byref(self.h)))
        return self

    def insert(self, cloud, T=None):
        t16 = None if T is None else _T16(T)
        check(load().sga_voxelmap_insert(self.ctx.h, self.h, cloud.h, None if t16 is None else _dp(t16)))

    def set_lru(self, horizon=100, clear_cycle=10):
        check(load().sga_voxelmap_set_lru(self.h, int(horizon), int(clear_cycle)))

    def set_search_offsets(self, num_offsets):
        """incremental_voxelmap.hpp:157-186: 1 (the query's own voxel), 7 or 27 voxels offer their Gaussians, the nearest mean wins."""
        check(load().sga_voxelmap_set_search_offsets(self.h, int(num_offsets)))

    def batch_knn_search(self, pts, k, max_sq_dist=-1.0):
        return _voxelmap_knn(self, pts, k, max_sq_dist)

    def knn_search(self, pt, k):
        idx, d2 = _voxelmap_knn(self, np.asarray(pt, dtype=np.float64).reshape(1, -1), k)
        return idx[0], d2[0]

    def __len__(self):
        return self.size()

    def size(self):
        if not self.h.value:
            return 0
        n = C.c_size_t()
        check(load().sga_index_size(self.h, C.byref(n)))
        return n.value

    def download(self):
        n = self.size()
        coords = np.empty((n, 3), np.int32)
        means = np.empty((n, 3), np.float32)
        c6 = np.empty((n, 6), np.float32)
        counts = np.empty(n, np.uint32)
        check(load().sga_index_voxelmap_download(self.ctx.h, self.h, coords.ctypes.data_as(C.POINTER(C.c_int32)), _fp(means), _fp(c6), counts.ctypes.data_as(C.POINTER(C.c_uint32))))
        return coords, means, c6, counts

    def voxel_points(self):
        m = self.download()[1].astype(np.float64)
        return np.concatenate([m, np.ones((len(m), 1))], axis=1)

    def voxel_covs(self):
        c6 = self.download()[2].astype(np.float64)
        out = np.zeros((len(c6), 4, 4))
        out[:, :3, :3] = mats_from_sym6(c6)
        return out


class IncrementalVoxelMapCov:
    """small_gicp.IncrementalVoxelMapCov (src/python/voxelmap.cpp:110-140) = IncrementalVoxelMap<FlatContainerCov>: voxels keep up
    to `max_num_points_in_cell` of the inserted points (with covariances); the scan-to-model GICP target.  `insert(cloud, T)`,
    `set_lru`, `set_search_offsets(1 | 7 | 27)`, `size()`, `voxel_points()`, `voxel_covs()`."""

    FLAT_CAP = 16

    def __init__(self, leaf_size, ctx=None):
        self.leaf = float(leaf_size)
        self.ctx = ctx or default_context()
        self.h = C.c_void_p()
        check(load().sga_flatmap_create(self.ctx.h, self.leaf, C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None) and self.h.value:
            load().sga_index_destroy(self.h)
            self.h = C.c_void_p()

    def insert(self, cloud, T=None):
        t16 = None if T is None else _T16(T)
        check(load().sga_voxelmap_insert(self.ctx.h, self.h, cloud.h, None if t16 is None else _dp(t16)))

    def set_lru(self, horizon=100, clear_cycle=10):
        check(load().sga_voxelmap_set_lru(self.h, int(horizon), int(clear_cycle)))

    def batch_knn_search(self, pts, k, max_sq_dist=-1.0):
        return _voxelmap_knn(self, pts, k, max_sq_dist)

    def knn_search(self, pt, k):
        idx, d2 = _voxelmap_knn(self, np.asarray(pt, dtype=np.float64).reshape(1, -1), k)
        return idx[0], d2[0]

    @classmethod
    def from_voxels(cls, leaf_size, coords, counts, points, cov6=None, search_offsets=1, ctx=None):
        """A map from voxels that exist on the host in the reference's flat order (sga_index_create_flatmap_from_voxels: what
        ParallelReductionHIP uploads for an IncrementalVoxelMap<FlatContainer*> target): coords (V, 3), counts (V,), points (P, 3) and
        cov6 (P, 6) with the points of voxel 0 first, then voxel 1, ... like download().  A search target only."""
        self = cls.__new__(cls)
        self.leaf = float(leaf_size)
        self.ctx = ctx or default_context()
        self.h = C.c_void_p()
        coords = np.ascontiguousarray(coords, dtype=np.int32).reshape(-1, 3)
        counts = np.ascontiguousarray(counts, dtype=np.uint32).reshape(-1)
        points = np.asarray(points, dtype=np.float64).reshape(-1, 3)
        n = len(coords)
        if len(counts) != n or int(counts.sum()) != len(points) or (n and counts.max() > cls.FLAT_CAP):
            raise ValueError("counts must have one entry per voxel (<= %d) and sum to the number of points" % cls.FLAT_CAP)
        valid = np.arange(cls.FLAT_CAP)[None, :] < counts[:, None]
        p16 = np.zeros((n, cls.FLAT_CAP, 3))
        p16[valid] = points
        c16 = None
        if cov6 is not None:
            c16 = np.zeros((n, cls.FLAT_CAP, 6))
            c16[valid] = np.asarray(cov6, dtype=np.float64).reshape(-1, 6)
        check(load().sga_index_create_flatmap_from_voxels(self.ctx.h, self.leaf, coords.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p), _dp(p16), None if c16 is None else _dp(c16),
                                                          int(search_offsets), n, C.byref(self.h)))
        return self

    def set_setting(self, min_sq_dist_in_cell=0.01, max_num_points_in_cell=10):
        check(load().sga_flatmap_set_setting(self.h, float(min_sq_dist_in_cell), int(max_num_points_in_cell)))

    def set_search_offsets(self, num_offsets):
        check(load().sga_voxelmap_set_search_offsets(self.h, int(num_offsets)))

    def size(self):
        n = C.c_size_t()
        check(load().sga_index_size(self.h, C.byref(n)))
        return n.value

    __len__ = size

    def download(self):
        """coords (V,3), counts (V,), points (P,3), cov6 (P,6): the points of voxel 0 first, then voxel 1, ... (P = sum of counts)."""
        n = self.size()
        coords = np.empty((n, 3), np.int32)
        counts = np.empty(n, np.uint32)
        pts = np.empty((n, self.FLAT_CAP, 3), np.float32)
        c6 = np.empty((n, self.FLAT_CAP, 6), np.float32)
        check(load().sga_flatmap_download(self.ctx.h, self.h, coords.ctypes.data_as(C.POINTER(C.c_int32)), counts.ctypes.data_as(C.POINTER(C.c_uint32)), _fp(pts), _fp(c6)))
        valid = np.arange(self.FLAT_CAP)[None, :] < counts[:, None]
        return coords, counts, pts[valid], c6[valid]

    def voxel_points(self):
        p = self.download()[2].astype(np.float64)
        return np.concatenate([p, np.ones((len(p), 1))], axis=1)

    def voxel_covs(self):
        c6 = self.download()[3].astype(np.float64)
        out = np.zeros((len(c6), 4, 4))
        out[:, :3, :3] = mats_from_sym6(c6)
        return out


class RegistrationResult:
    """registration_result.hpp:11-30, field for field."""

    def __init__(self, rc=None):
        if rc is None:
            self.T_target_source = np.eye(4)
            self.converged, self.iterations, self.num_inliers = False, 0, 0
            self.H, self.b, self.error = np.zeros((6, 6)), np.zeros(6), 0.0
        else:
            self.T_target_source = np.array(rc.T_target_source).reshape(4, 4).T.copy()
            self.converged = bool(rc.converged)
            self.iterations = int(rc.iterations)
            self.num_inliers = int(rc.num_inliers)
            self.H = np.array(rc.H).reshape(6, 6)
            self.b = np.array(rc.b)
            self.error = float(rc.error)

    def __repr__(self):
        return f"RegistrationResult(converged={self.converged}, iterations={self.iterations}, num_inliers={self.num_inliers}, error={self.error:.6g})"


def make_setting(
    registration_type="GICP",
    max_correspondence_distance=1.0,
    max_iterations=20,
    rotation_eps=0.1 * np.pi / 180.0,
    translation_eps=1e-3,
    robust_kernel=None,
    robust_c=1.0,
    optimizer="LM",
    math_mode="fp32",
    verbose=False,
    max_inner_iterations=10,
    init_lambda=1e-3,
    lambda_factor=10.0,
    gn_lambda=1e-6,
    restrict_dof_lambda=0.0,
    restrict_dof_mask=None,
):
    s = RegistrationSettingC()
    load().sga_registration_setting_default(C.byref(s))
    s.factor.factor_kind = _FACTOR_BY_NAME[registration_type] if isinstance(registration_type, str) else int(registration_type)
    # None: no rejector (NullRejector, rejector.hpp:11-16).  Any number, negative ones included, is squared like the reference does
    # (registration_helper.cpp:90,100,110; src/python/align.cpp:246): DistanceRejector with max_dist_sq = d * d
    s.factor.max_dist_sq = -1.0 if max_correspondence_distance is None else float(max_correspondence_distance) ** 2
    s.factor.robust_kind = {None: 0, "NONE": 0, "HUBER": 1, "CAUCHY": 2}[robust_kernel if robust_kernel is None else robust_kernel.upper()]
    s.factor.robust_c = float(robust_c)
    s.factor.math_mode = {"fp32": 0, "fp64": 1}[math_mode]
    s.optimizer = {"LM": 0, "GN": 1}[optimizer]
    s.max_iterations = int(max_iterations)
    s.max_inner_iterations = int(max_inner_iterations)
    s.init_lambda, s.lambda_factor, s.gn_lambda = float(init_lambda), float(lambda_factor), float(gn_lambda)
    s.rotation_eps, s.translation_eps = float(rotation_eps), float(translation_eps)
    s.verbose = int(verbose)
    s.restrict_dof_lambda = float(restrict_dof_lambda)
    if restrict_dof_mask is not None:
        for i in range(6):
            s.restrict_dof_mask[i] = float(restrict_dof_mask[i])
    return s


class Problem:
    """(target index, source cloud) pairing with device-resident factor state: the Reduction slot of Registration<>."""

    def __init__(self, target, source, init_T=None, ctx=None):
        self.target, self.source = target, source
        self.ctx = ctx or source.ctx  # a problem may run on another context (stream) of the same device than the one that built its inputs
        self.h = C.c_void_p()
        t16 = _T16(init_T)
        if isinstance(source, KdTree):  # the source by its own index: its kd order is taken as it is (no sort)
            check(load().sga_problem_create_from_index(self.ctx.h, target.h, source.h, _dp(t16), C.byref(self.h)))
        else:
            check(load().sga_problem_create(self.ctx.h, target.h, source.h, _dp(t16), C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None) and self.h.value:
            load().sga_problem_destroy(self.h)
            self.h = C.c_void_p()

    def linearize(self, factor_params, T):
        H, b = np.empty(36), np.empty(6)
        e, ninl = C.c_double(), C.c_uint64()
        t16 = _T16(T)
        check(load().sga_linearize(self.ctx.h, self.h, C.byref(factor_params), _dp(t16), _dp(H), _dp(b), C.byref(e), C.byref(ninl)))
        return H.reshape(6, 6), b, e.value, ninl.value

    def error(self, factor_params, T):
        e = C.c_double()
        t16 = _T16(T)
        check(load().sga_error(self.ctx.h, self.h, C.byref(factor_params), _dp(t16), C.byref(e)))
        return e.value

    def linearize_async(self, factor_params, T, d_out_ptr):
        t16 = _T16(T)
        check(load().sga_linearize_async(self.ctx.h, self.h, C.byref(factor_params), _dp(t16), C.c_void_p(int(d_out_ptr))))

    def error_async(self, factor_params, T, d_out_ptr):
        t16 = _T16(T)
        check(load().sga_error_async(self.ctx.h, self.h, C.byref(factor_params), _dp(t16), C.c_void_p(int(d_out_ptr))))

    def set_rejector(self, fn):
        """A custom CorrespondenceRejector (rejector.hpp:11-28) as a batch callback: fn(T 4x4, target_index (n,) int64, sq_dist (n,)
        float32) -> boolean array, True = reject.  None restores the built-in distance rejector."""
        if fn is None:
            self._rejector_cb = _lib.REJECTOR_FN()  # NULL function pointer
        else:
            def cb(_user, T16, n, idx_p, d2_p, rej_p):
                try:
                    T = np.ctypeslib.as_array(T16, (16,)).reshape(4, 4).T.copy()
                    idx = np.ctypeslib.as_array(idx_p, (n,))
                    d2 = np.ctypeslib.as_array(d2_p, (n,))
                    out = np.ctypeslib.as_array(rej_p, (n,))
                    out[:] = np.asarray(fn(T, idx, d2), dtype=bool)
                    return 0
                except Exception:  # noqa: BLE001
                    import traceback

                    traceback.print_exc()
                    return 1

            self._rejector_cb = _lib.REJECTOR_FN(cb)
        check(load().sga_problem_set_rejector(self.h, self._rejector_cb, None))

    def linearize_per_point(self, factor_params, T):
        """sga_linearize_per_point: (inlier (n,) bool, H (n,6,6), b (n,6), e (n,)) for every source point in the caller's order."""
        n = self.source.size()
        vals = np.zeros((n, 28))
        ok = np.zeros(n, np.uint8)
        t16 = _T16(T)
        check(load().sga_linearize_per_point(self.ctx.h, self.h, C.byref(factor_params), _dp(t16), _dp(vals), ok.ctypes.data_as(C.POINTER(C.c_ubyte))))
        H = np.zeros((n, 6, 6))
        iu = np.triu_indices(6)
        H[:, iu[0], iu[1]] = vals[:, :21]
        H[:, iu[1], iu[0]] = vals[:, :21]
        return ok.astype(bool), H, vals[:, 21:27].copy(), vals[:, 27].copy()

    def factors(self):
        n = self.source.size()
        ti = np.empty(n, np.int64)
        m6 = np.empty((n, 6), np.float32)
        check(load().sga_problem_get_factors(self.ctx.h, self.h, ti.ctypes.data_as(C.POINTER(C.c_int64)), _fp(m6)))
        return ti, m6

    def align(self, setting, init_T=None):
        res = ResultC()
        t16 = _T16(init_T)
        check(load().sga_align_problem(self.ctx.h, self.h, _dp(t16), C.byref(setting), C.byref(res)))
        return RegistrationResult(res)

    def search_stats(self, enable=None):
        """Diagnostics: enable / disable the per-point record of leaves scanned, or (enable=None) fetch the last pass's counts."""
        if enable is not None:
            check(load().sga_problem_set_search_stats(self.ctx.h, self.h, 1 if enable else 0))
            return None
        out = np.zeros(len(self.source), dtype=np.int32)
        check(load().sga_problem_get_search_stats(self.ctx.h, self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def sorted_points(self):
        """Diagnostics: the source points in the engine's order (n x 4 float32; the 4th column holds the original index as bits)."""
        out = np.zeros((len(self.source), 4), dtype=np.float32)
        check(load().sga_problem_get_sorted_points(self.ctx.h, self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def pass_stats(self):
        """Linearization passes since creation by kind (cold = full search, warm = certified neighbours) and the source points
        the warm passes had to search again."""
        c, w, f = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(load().sga_problem_get_pass_stats(self.ctx.h, self.h, C.byref(c), C.byref(w), C.byref(f)))
        g = (C.c_uint64 * 6)()
        check(load().sga_problem_get_grid_stats(self.h, g))
        return {"cold_passes": c.value, "warm_passes": w.value, "walked_points": f.value, "grid_passes": g[0], "grid_open": g[1], "grid_rings": g[2], "grid_cell_m": g[3] * 1e-6,
                "adj_queries": g[4], "adj_unsettled": g[5]}


class MultiProblem:
    """sga_multi: one registration over several GPUs of this process (source sharded, target replicated; small_gicp_amd.h).  `devices`
    may name a device more than once (logical shards on one GPU).  Clouds are given as host arrays: points (n, 3), normals (n, 3) or
    None, covariances (n, 3, 3) or None."""

    def __init__(self, devices, target, source, init_T=None):
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        self.h = C.c_void_p()
        check(load().sga_multi_create(devs, len(devices), C.byref(self.h)))
        tp, tn, tc = self._pack(*target)
        check(load().sga_multi_set_target_f64(self.h, _dp(tp), _dp(tn) if tn is not None else None, _dp(tc) if tc is not None else None, len(tp)))
        sp, sn, sc = self._pack(*source)
        self.n_source = len(sp)
        check(load().sga_multi_set_source_f64(self.h, _dp(sp), _dp(sn) if sn is not None else None, _dp(sc) if sc is not None else None, len(sp), _dp(_T16(np.eye(4) if init_T is None else init_T))))

    @staticmethod
    def _pack(points, normals=None, covs=None):
        p = np.ones((len(points), 4))
        p[:, :3] = np.asarray(points, dtype=np.float64)[:, :3]
        nr = cv = None
        if normals is not None:
            nr = np.zeros((len(points), 4))
            nr[:, :3] = np.asarray(normals, dtype=np.float64)[:, :3]
        if covs is not None:
            cv = np.zeros((len(points), 4, 4))
            cv[:, :3, :3] = np.asarray(covs, dtype=np.float64)[:, :3, :3]
            cv = np.ascontiguousarray(cv.transpose(0, 2, 1))  # column-major 4x4 (symmetric: the same numbers)
        return np.ascontiguousarray(p), nr, cv

    def __del__(self):
        if getattr(self, "h", None):
            load().sga_multi_destroy(self.h)
            self.h = None

    def linearize(self, factor_params, T):
        H, b, e, n = np.zeros(36), np.zeros(6), C.c_double(), C.c_uint64()
        check(load().sga_multi_linearize(self.h, C.byref(factor_params), _dp(_T16(T)), _dp(H), _dp(b), C.byref(e), C.byref(n)))
        return H.reshape(6, 6), b, e.value, n.value

    def error(self, factor_params, T):
        e = C.c_double()
        check(load().sga_multi_error(self.h, C.byref(factor_params), _dp(_T16(T)), C.byref(e)))
        return e.value

    def align(self, setting, init_T=None):
        res = ResultC()
        check(load().sga_multi_align(self.h, _dp(_T16(np.eye(4) if init_T is None else init_T)), C.byref(setting), C.byref(res)))
        return RegistrationResult(res)

    def factors(self):
        idx = np.empty(self.n_source, np.int64)
        check(load().sga_multi_get_factors(self.h, idx.ctypes.data_as(C.c_void_p), None))
        return idx


def _voxelmap_knn(vm, pts, k, max_sq_dist=-1.0):
    """traits::knn_search of a voxel map (incremental_voxelmap.hpp:127-149) for m queries: global indices (voxel_id << 32) | point_id
    (-1 = none) and squared distances ascending (inf = none), each (m, k)."""
    q = np.ascontiguousarray(np.asarray(pts, dtype=np.float64).reshape(-1, np.asarray(pts).shape[-1])[:, :3])
    idx = np.empty((len(q), k), np.int64)
    d2 = np.empty((len(q), k), np.float64)
    check(load().sga_index_knn_f64(vm.ctx.h, vm.h, _dp(q), len(q), int(k), float(max_sq_dist), idx.ctypes.data_as(C.POINTER(C.c_int64)), _dp(d2)))
    return idx, d2


def set_warm_limit(warm_delta_m):
    """sga_set_warm_limit: a negative limit makes every linearization pass walk in full (used by tests to compare the two kinds of pass)."""
    load().sga_set_warm_limit(float(warm_delta_m))


def set_error_model(enabled):
    """sga_set_error_model: False makes every error pass run the error kernel instead of the quadratic model of the last linearization."""
    load().sga_set_error_model(1 if enabled else 0)


def set_search_mode(queue=2, chunk_tiles_cold=0, chunk_tiles_warm=0):
    """sga_set_search_mode: 1 / True = queue-fed search kernel, 0 / False = one query per lane, 2 = automatic (default); tiles of 64
    source points per wave of the queue-fed kernel."""
    load().sga_set_search_mode(int(queue), int(chunk_tiles_cold), int(chunk_tiles_warm))


def set_grid_mode(mode=-1, min_points=-1):
    """sga_set_grid_mode (small_gicp_amd_debug.h): what the cell grid is used for (0 nothing, 1 default: the walkers of warm passes, 2 also
    cold passes but a registration's first, 3 the first too, 4 every pass) and from how many target points on an index gets one.  Results
    do not depend on it."""
    load().sga_set_grid_mode(int(mode), int(min_points))


def get_warm_limit():
    return float(load().sga_get_warm_limit())


def unpack_accumulator(acc30):
    a = np.ascontiguousarray(acc30, dtype=np.float64)
    H, b = np.empty(36), np.empty(6)
    e, n = C.c_double(), C.c_uint64()
    load().sga_unpack_accumulator(_dp(a), _dp(H), _dp(b), C.byref(e), C.byref(n))
    return H.reshape(6, 6), b, e.value, n.value


def error_model_eval(acc96, T_lin, T):
    """sga_error_model_eval (host only): the error at trial pose T from the 96-double accumulator of a linearization at T_lin."""
    a = np.ascontiguousarray(acc96, dtype=np.float64)
    assert a.size >= 96
    tl, t = _T16(T_lin), _T16(T)
    e = C.c_double()
    check(load().sga_error_model_eval(_dp(a), _dp(tl), _dp(t), C.byref(e)))
    return e.value


def optimize(setting, init_T, linearize, error):
    """sga_optimize: the host LM/GN over python callbacks linearize(T)->(H,b,e,num_inliers), error(T)->e.  T is 4x4 row-major numpy."""

    def _lin(user, T, H, b, e, n):
        try:
            Tm = np.ctypeslib.as_array(T, shape=(16,)).reshape(4, 4).T
            h, bb, ee, nn = linearize(Tm.copy())
            np.ctypeslib.as_array(H, shape=(36,))[:] = np.asarray(h, dtype=np.float64).reshape(36)
            np.ctypeslib.as_array(b, shape=(6,))[:] = np.asarray(bb, dtype=np.float64).reshape(6)
            e[0] = float(ee)
            n[0] = int(nn)
            return 0
        except Exception as ex:  # noqa: BLE001
            _lin.exc = ex
            return 1

    def _err(user, T, e):
        try:
            Tm = np.ctypeslib.as_array(T, shape=(16,)).reshape(4, 4).T
            e[0] = float(error(Tm.copy()))
            return 0
        except Exception as ex:  # noqa: BLE001
            _lin.exc = ex
            return 1

    _lin.exc = None
    res = ResultC()
    t16 = _T16(init_T)
    rc = load().sga_optimize(C.byref(setting), _dp(t16), _lib.LINEARIZE_FN(_lin), _lib.ERROR_FN(_err), None, C.byref(res))
    if _lin.exc is not None:
        raise _lin.exc
    check(rc)
    return RegistrationResult(res)


# ---- preprocessing (src/python/preprocess.cpp) -----------------------------------------------------------------------------
def voxelgrid_sampling(points, downsampling_resolution, num_threads=1):
    cloud = points if isinstance(points, PointCloud) else PointCloud(points)
    out = C.c_void_p()
    check(load().sga_voxelgrid_sampling(cloud.ctx.h, cloud.h, float(downsampling_resolution), C.byref(out)))
    return PointCloud(ctx=cloud.ctx, _handle=out)


def _estimate(cloud, tree, num_neighbors, flags):
    check(load().sga_estimate_normals_covariances(cloud.ctx.h, cloud.h, tree.h if tree is not None else None, int(num_neighbors), flags))


def estimate_normals(points, tree=None, num_neighbors=20, num_threads=1):
    _estimate(points, tree, num_neighbors, 1)


def estimate_covariances(points, tree=None, num_neighbors=20, num_threads=1):
    _estimate(points, tree, num_neighbors, 2)


def estimate_normals_covariances(points, tree=None, num_neighbors=20, num_threads=1):
    _estimate(points, tree, num_neighbors, 3)


def preprocess_points(points, downsampling_resolution=0.25, num_neighbors=10, num_threads=1):
    """registration_helper.cpp:22-34: downsample -> index -> normals + covariances.  Returns (PointCloud, KdTree)."""
    cloud = points if isinstance(points, PointCloud) else PointCloud(points)
    down = voxelgrid_sampling(cloud, downsampling_resolution)
    tree = KdTree(down)  # (built first: the estimation searches it and fills its kd-ordered attribute copies — one build, not two)
    estimate_normals_covariances(down, tree, num_neighbors)
    return down, tree


def align(
    target,
    source,
    target_tree=None,
    init_T_target_source=None,
    registration_type="GICP",
    voxel_resolution=1.0,
    downsampling_resolution=0.25,
    max_correspondence_distance=1.0,
    num_threads=1,
    max_iterations=20,
    verbose=False,
    **kw,
):
    """small_gicp.align (src/python/align.cpp:22-295), three call forms:
    align(target_numpy, source_numpy, ...)            -> preprocess both, then register          (registration_helper.cpp:57-69)
    align(target_cloud, source_cloud, target_tree)    -> ICP / PLANE_ICP / GICP                    (registration_helper.cpp:81-122)
    align(target_voxelmap, source_cloud)              -> VGICP                                      (registration_helper.cpp:125-137)
    """
    if isinstance(target, GaussianVoxelMap):
        # the Python binding's voxel-map overload DOES apply max_correspondence_distance (src/python/align.cpp:246), unlike the C++
        # helper align(GaussianVoxelMap, ...) of registration_helper.cpp:125-137 (mirrored in include/small_gicp_amd.hpp), which
        # leaves the rejector at 1.0 m^2: each layer mirrors its own counterpart
        setting = make_setting("GICP", max_correspondence_distance, max_iterations, verbose=verbose, **kw)
        return Problem(target, source, init_T_target_source).align(setting, init_T_target_source)
    if not isinstance(target, PointCloud):
        tgt, tree = preprocess_points(np.asarray(target), downsampling_resolution, 10)
        src, _ = preprocess_points(np.asarray(source), downsampling_resolution, 10)
        if registration_type == "VGICP":
            vm = GaussianVoxelMap(voxel_resolution, ctx=tgt.ctx)
            vm.insert(tgt)
            # registration_helper.cpp:130-136 leaves the rejector at its default 1.0 m^2 for VGICP
            setting = make_setting("GICP", 1.0, max_iterations, verbose=verbose, **kw)
            return Problem(vm, src, init_T_target_source).align(setting, init_T_target_source)
        target, source, target_tree = tgt, src, tree
    if target_tree is None:
        target_tree = KdTree(target)
    target_tree.refresh_attributes()
    setting = make_setting(registration_type, max_correspondence_distance, max_iterations, verbose=verbose, **kw)
    return Problem(target_tree, source, init_T_target_source).align(setting, init_T_target_source)
