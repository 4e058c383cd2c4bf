"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

ctypes driver for oracle/_ref/libsmall_gicp_ref.so: the UNMODIFIED reference (koide3/small_gicp v1.0.1 headers +
registration_helper.cpp, compiled in place from /root/reference by oracle/ref/Makefile against the home-made Eigen stand-in
oracle/ref/eigen_shim).  It exists to validate the CPU restatement in oracle/ against the reference's own code and, where
present, to serve as the timed CPU baseline (kind "reference").  The .so is git-ignored and only buildable where /root/reference
is mounted; it travels to the GPU box with the snapshot like other built libraries.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libsmall_gicp_ref.so")
_LIB = None

ICP, PLANE_ICP, GICP, VGICP = 0, 1, 2, 3


class Result(C.Structure):
    _fields_ = [("T", C.c_double * 16), ("converged", C.c_int), ("iterations", C.c_uint64), ("num_inliers", C.c_uint64), ("H", C.c_double * 36), ("b", C.c_double * 6), ("error", C.c_double)]


LIB_PATH_V3 = os.path.join(_HERE, "_ref", "libsmall_gicp_ref_v3.so")  # the same sources compiled with -march=x86-64-v3


def available():
    return os.path.exists(LIB_PATH)


def select(variant=None):
    """Switch to another build of the reference ("v3": -march=x86-64-v3; None: the default -O3 build).  Objects created before the
    switch belong to the library that created them: destroy them first.  Returns False if that build is not there."""
    global _LIB, LIB_PATH
    path = {None: os.path.join(_HERE, "_ref", "libsmall_gicp_ref.so"), "v3": LIB_PATH_V3}[variant]
    if not os.path.exists(path):
        return False
    if path != LIB_PATH:
        LIB_PATH, _LIB = path, None
    return True


def build():
    """Compile the reference in place (needs /root/reference). Returns True if the library exists afterwards."""
    if os.path.isdir("/root/reference/include/small_gicp"):
        subprocess.call(["make", "-C", os.path.join(_HERE, "ref")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return available()


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(LIB_PATH)
        vp, dp = C.c_void_p, C.POINTER(C.c_double)
        L.ref_cloud_create.argtypes = [dp, dp, dp, C.c_size_t, C.c_int, C.c_int]
        L.ref_cloud_create.restype = vp
        L.ref_cloud_destroy.argtypes = [vp]
        L.ref_cloud_size.argtypes = [vp]
        L.ref_cloud_size.restype = C.c_size_t
        L.ref_cloud_get.argtypes = [vp, dp, dp, dp]
        L.ref_voxelgrid_sampling.argtypes = [vp, C.c_double]
        L.ref_voxelgrid_sampling.restype = vp
        L.ref_estimate_normals_covariances.argtypes = [vp, C.c_int, C.c_int]
        L.ref_estimate_covariances.argtypes = [vp, C.c_int, C.c_int]
        L.ref_nearest.argtypes = [vp, dp, C.c_size_t, C.c_int, C.POINTER(C.c_int64), dp]
        L.ref_knn.argtypes = [vp, dp, C.c_size_t, C.c_int, C.POINTER(C.c_int64), dp]
        L.ref_knn.restype = C.c_size_t
        L.ref_align.argtypes = [vp, vp, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, C.c_double, dp, C.POINTER(Result), dp]
        L.ref_align_general.argtypes = [vp, vp, C.c_int, C.c_double, dp, C.c_double, C.c_int, C.c_int, C.c_double, C.c_double, dp, C.POINTER(Result)]
        L.ref_linearize.argtypes = [vp, vp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, dp, dp, dp, dp, dp, C.POINTER(C.c_uint64)]
        L.ref_voxelmap_size.argtypes = [vp, C.c_double]
        L.ref_voxelmap_size.restype = C.c_size_t
        L.ref_ivm_create.argtypes = [C.c_double]
        L.ref_ivm_create.restype = vp
        L.ref_ivm_destroy.argtypes = [vp]
        L.ref_ivm_set_lru.argtypes = [vp, C.c_size_t, C.c_size_t]
        L.ref_ivm_insert.argtypes = [vp, vp, dp]
        L.ref_ivm_size.argtypes = [vp]
        L.ref_ivm_size.restype = C.c_size_t
        L.ref_ivm_get.argtypes = [vp, C.POINTER(C.c_int), dp, dp, C.POINTER(C.c_uint64)]
        L.ref_fvm_create.argtypes = [C.c_double]
        L.ref_fvm_create.restype = vp
        L.ref_fvm_destroy.argtypes = [vp]
        L.ref_fvm_set_lru.argtypes = [vp, C.c_size_t, C.c_size_t]
        L.ref_fvm_set_setting.argtypes = [vp, C.c_double, C.c_size_t]
        L.ref_fvm_set_search_offsets.argtypes = [vp, C.c_int]
        L.ref_fvm_insert.argtypes = [vp, vp, dp]
        L.ref_fvm_size.argtypes = [vp]
        L.ref_fvm_size.restype = C.c_size_t
        L.ref_fvm_total_points.argtypes = [vp]
        L.ref_fvm_total_points.restype = C.c_size_t
        L.ref_fvm_get.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_uint64), dp, dp]
        L.ref_fvm_align.argtypes = [vp, vp, C.c_int, dp, C.POINTER(Result)]
        L.ref_fvm_knn.argtypes = [vp, dp, C.c_size_t, C.c_int, C.POINTER(C.c_int64), dp]
        L.ref_ivm_knn.argtypes = [vp, dp, C.c_size_t, C.c_int, C.POINTER(C.c_int64), dp]
        _LIB = L
    return _LIB


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a, last):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1, last))


def _T16(T):
    return np.ascontiguousarray(np.asarray(T, dtype=np.float64).reshape(4, 4).T).reshape(16)


class Cloud:
    def __init__(self, points=None, normals=None, covs=None, tree=True, tree_threads=1, _handle=None):
        if _handle is not None:
            self.h = _handle
            return
        p = _f64(points, 3)
        n = _f64(normals, 3)
        c = None if covs is None else _f64(np.asarray(covs).reshape(len(p), 9), 9)
        self.h = lib().ref_cloud_create(_dp(p), _dp(n), _dp(c), len(p), int(tree), int(tree_threads))

    def __del__(self):
        if getattr(self, "h", None):
            lib().ref_cloud_destroy(self.h)
            self.h = None

    def __len__(self):
        return lib().ref_cloud_size(self.h)

    def get(self):
        n = len(self)
        p, nr, cv = np.empty((n, 3)), np.empty((n, 3)), np.empty((n, 9))
        lib().ref_cloud_get(self.h, _dp(p), _dp(nr), _dp(cv))
        return p, nr, cv.reshape(n, 3, 3)

    def voxelgrid_sampling(self, leaf):
        return Cloud(_handle=lib().ref_voxelgrid_sampling(self.h, float(leaf)))

    def estimate_normals_covariances(self, k=20, num_threads=1):
        lib().ref_estimate_normals_covariances(self.h, int(k), int(num_threads))

    def estimate_covariances(self, k=20, num_threads=1):
        lib().ref_estimate_covariances(self.h, int(k), int(num_threads))

    def nearest(self, queries, num_threads=1):
        """KdTree::nearest_neighbor_search per query (double): (index or -1, squared distance)."""
        q = _f64(queries, 3)
        idx = np.empty(len(q), np.int64)
        d2 = np.empty(len(q))
        lib().ref_nearest(self.h, _dp(q), len(q), int(num_threads), idx.ctypes.data_as(C.POINTER(C.c_int64)), _dp(d2))
        return idx, d2

    def knn(self, queries, k):
        q = _f64(queries, 3)
        idx = np.empty((len(q), k), np.int64)
        d2 = np.empty((len(q), k))
        lib().ref_knn(self.h, _dp(q), len(q), int(k), idx.ctypes.data_as(C.POINTER(C.c_int64)), _dp(d2))
        return idx, d2

    def voxelmap_size(self, leaf):
        return lib().ref_voxelmap_size(self.h, float(leaf))


class VoxelMap:
    """The reference's GaussianVoxelMap used incrementally: insert(cloud, T) any number of times (incremental_voxelmap.hpp:55-92)."""

    def __init__(self, leaf):
        self.h = lib().ref_ivm_create(float(leaf))

    def __del__(self):
        if getattr(self, "h", None):
            lib().ref_ivm_destroy(self.h)
            self.h = None

    def set_lru(self, horizon=100, clear_cycle=10):
        lib().ref_ivm_set_lru(self.h, int(horizon), int(clear_cycle))

    def insert(self, cloud, T=None):
        t16 = None if T is None else _T16(T)
        lib().ref_ivm_insert(self.h, cloud.h, _dp(t16))

    def __len__(self):
        return lib().ref_ivm_size(self.h)

    def knn(self, queries, k):
        return _map_knn(lib().ref_ivm_knn, self.h, queries, k)

    def get(self):
        n = len(self)
        coords = np.empty((n, 3), np.int32)
        means = np.empty((n, 3))
        covs = np.empty((n, 9))
        counts = np.empty(n, np.uint64)
        lib().ref_ivm_get(self.h, coords.ctypes.data_as(C.POINTER(C.c_int)), _dp(means), _dp(covs), counts.ctypes.data_as(C.POINTER(C.c_uint64)))
        return coords, means, covs.reshape(n, 3, 3), counts


def _map_knn(fn, h, queries, k):
    q = _f64(queries, 3)
    idx = np.empty((len(q), k), np.int64)
    d2 = np.empty((len(q), k))
    fn(h, _dp(q), len(q), int(k), idx.ctypes.data_as(C.POINTER(C.c_int64)), _dp(d2))
    return idx, d2


class FlatMap:
    """The reference's IncrementalVoxelMap<FlatContainerCov> (scan-to-model GICP target)."""

    def __init__(self, leaf):
        self.h = lib().ref_fvm_create(float(leaf))

    def __del__(self):
        if getattr(self, "h", None):
            lib().ref_fvm_destroy(self.h)
            self.h = None

    def set_lru(self, horizon=100, clear_cycle=10):
        lib().ref_fvm_set_lru(self.h, int(horizon), int(clear_cycle))

    def set_setting(self, min_sq_dist_in_cell=0.01, max_num_points_in_cell=10):
        lib().ref_fvm_set_setting(self.h, float(min_sq_dist_in_cell), int(max_num_points_in_cell))

    def set_search_offsets(self, n):
        lib().ref_fvm_set_search_offsets(self.h, int(n))

    def insert(self, cloud, T=None):
        t16 = None if T is None else _T16(T)
        lib().ref_fvm_insert(self.h, cloud.h, _dp(t16))

    def __len__(self):
        return lib().ref_fvm_size(self.h)

    def get(self):
        n, total = len(self), lib().ref_fvm_total_points(self.h)
        coords = np.empty((n, 3), np.int32)
        counts = np.empty(n, np.uint64)
        pts = np.empty((total, 3))
        covs = np.empty((total, 9))
        lib().ref_fvm_get(self.h, coords.ctypes.data_as(C.POINTER(C.c_int)), counts.ctypes.data_as(C.POINTER(C.c_uint64)), _dp(pts), _dp(covs))
        return coords, counts, pts, covs.reshape(total, 3, 3)

    def knn(self, queries, k):
        """IncrementalVoxelMap::knn_search per query: (global indices (voxel << 32) | point or -1, squared distances or inf)."""
        return _map_knn(lib().ref_fvm_knn, self.h, queries, k)

    def align(self, source, init_T=None, num_threads=4):
        res = Result()
        t16 = _T16(np.eye(4) if init_T is None else init_T)
        assert lib().ref_fvm_align(self.h, source.h, int(num_threads), _dp(t16), C.byref(res)) == 0
        return _result(res)


class AlignResult:
    pass


def align(target, source, type=GICP, voxel_resolution=1.0, max_correspondence_distance=1.0, num_threads=4, max_iterations=20, rotation_eps=0.1 * np.pi / 180.0, translation_eps=1e-3, init_T=None):
    res = Result()
    el = C.c_double()
    t16 = _T16(np.eye(4) if init_T is None else init_T)
    rc = lib().ref_align(target.h, source.h, int(type), float(voxel_resolution), float(max_correspondence_distance), int(num_threads), int(max_iterations), float(rotation_eps), float(translation_eps), _dp(t16), C.byref(res), C.byref(el))
    assert rc == 0
    r = _result(res)
    r.elapsed_sec = el.value
    return r


def align_general(target, source, optimizer=0, restrict_lambda=0.0, restrict_mask=(1, 1, 1, 1, 1, 1), max_correspondence_distance=1.0, num_threads=4, max_iterations=20, rotation_eps=0.1 * np.pi / 180.0,
                  translation_eps=1e-3, init_T=None):
    """Registration<GICPFactor, ParallelReductionOMP, RestrictDoFFactor | NullFactor, DistanceRejector, LM (0) | GN (1)>::align."""
    res = Result()
    t16 = _T16(np.eye(4) if init_T is None else init_T)
    m = np.ascontiguousarray(restrict_mask, dtype=np.float64)
    rc = lib().ref_align_general(target.h, source.h, int(optimizer), float(restrict_lambda), _dp(m), float(max_correspondence_distance), int(num_threads), int(max_iterations), float(rotation_eps), float(translation_eps),
                                 _dp(t16), C.byref(res))
    assert rc == 0
    return _result(res)


def _result(res):
    r = AlignResult()
    r.T_target_source = np.array(res.T).reshape(4, 4).T.copy()
    r.converged, r.iterations, r.num_inliers = bool(res.converged), int(res.iterations), int(res.num_inliers)
    r.H, r.b, r.error = np.array(res.H).reshape(6, 6), np.array(res.b), float(res.error)
    return r


def linearize(target, source, type=GICP, robust=0, robust_c=1.0, max_dist_sq=1.0, num_threads=1, T=None):
    H, b = np.empty(36), np.empty(6)
    e, e2 = C.c_double(), C.c_double()
    n = C.c_uint64()
    t16 = _T16(np.eye(4) if T is None else T)
    rc = lib().ref_linearize(target.h, source.h, int(type), int(robust), float(robust_c), float(max_dist_sq), int(num_threads), _dp(t16), _dp(H), _dp(b), C.byref(e), C.byref(e2), C.byref(n))
    assert rc == 0
    return H.reshape(6, 6), b, e.value, e2.value, n.value
