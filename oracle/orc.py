"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

ctypes driver for the CPU oracle (oracle/liboracle.so, built by oracle/Makefile from oracle.hpp — a double-precision
restatement of the small_gicp hot path; see oracle.hpp for reference citations and the pinning statement).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ICP, PLANE_ICP, GICP = 0, 1, 2
ROBUST_NONE, ROBUST_HUBER, ROBUST_CAUCHY = 0, 1, 2


class Setting(C.Structure):
    _fields_ = [
        ("factor_kind", C.c_int),
        ("robust_kind", C.c_int),
        ("robust_c", C.c_double),
        ("max_dist_sq", C.c_double),
        ("num_threads", C.c_int),
        ("optimizer_type", C.c_int),
        ("max_iterations", C.c_int),
        ("max_inner_iterations", C.c_int),
        ("init_lambda", C.c_double),
        ("lambda_factor", C.c_double),
        ("gn_lambda", C.c_double),
        ("translation_eps", C.c_double),
        ("rotation_eps", C.c_double),
        ("verbose", C.c_int),
        ("restrict_lambda", C.c_double),
        ("restrict_mask", C.c_double * 6),
    ]


class Result(C.Structure):
    _fields_ = [
        ("T", C.c_double * 16),
        ("converged", C.c_int),
        ("iterations", C.c_uint64),
        ("num_inliers", C.c_uint64),
        ("H", C.c_double * 36),
        ("b", C.c_double * 6),
        ("error", C.c_double),
    ]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("oracle_capi.cpp", "oracle.hpp", "orc_math.hpp")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs if os.path.exists(s)):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)
        L.orc_default_setting.argtypes = [C.POINTER(Setting)]
        L.orc_fast_floor.argtypes = [C.c_double]
        L.orc_fast_floor.restype = C.c_int
        L.orc_voxelgrid_sampling.argtypes = [dp, C.c_size_t, C.c_double, dp]
        L.orc_voxelgrid_sampling.restype = C.c_size_t
        L.orc_cloud_create.argtypes = [dp, dp, dp, C.c_size_t, C.c_int]
        L.orc_cloud_create.restype = vp
        L.orc_cloud_destroy.argtypes = [vp]
        L.orc_cloud_size.argtypes = [vp]
        L.orc_cloud_size.restype = C.c_size_t
        L.orc_cloud_get.argtypes = [vp, dp, dp, dp]
        L.orc_knn.argtypes = [vp, dp, C.c_size_t, C.c_int, ip, dp, C.c_int]
        L.orc_knn.restype = C.c_size_t
        L.orc_estimate_normals_covariances.argtypes = [vp, C.c_int, C.c_int]
        L.orc_eigen_sym3.argtypes = [dp, C.c_int, dp, dp]
        L.orc_se3_exp.argtypes = [dp, dp]
        L.orc_ldlt_solve.argtypes = [dp, dp, dp]
        L.orc_voxelmap_create.argtypes = [vp, C.c_double]
        L.orc_voxelmap_create.restype = vp
        L.orc_voxelmap_destroy.argtypes = [vp]
        L.orc_voxelmap_new.argtypes = [C.c_double]
        L.orc_voxelmap_new.restype = vp
        L.orc_voxelmap_insert.argtypes = [vp, vp, dp]
        L.orc_voxelmap_set_lru.argtypes = [vp, C.c_size_t, C.c_size_t]
        L.orc_flatmap_new.argtypes = [C.c_double]
        L.orc_flatmap_new.restype = vp
        L.orc_flatmap_set_setting.argtypes = [vp, C.c_double, C.c_size_t]
        L.orc_flatmap_total_points.argtypes = [vp]
        L.orc_flatmap_total_points.restype = C.c_size_t
        L.orc_flatmap_get.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_uint64), dp, dp]
        L.orc_voxelmap_size.argtypes = [vp]
        L.orc_voxelmap_size.restype = C.c_size_t
        L.orc_voxelmap_set_search_offsets.argtypes = [vp, C.c_int]
        L.orc_voxelmap_get.argtypes = [vp, C.POINTER(C.c_int), dp, dp, C.POINTER(C.c_uint64)]
        L.orc_factors_create.argtypes = [C.c_size_t]
        L.orc_factors_create.restype = vp
        L.orc_factors_destroy.argtypes = [vp]
        L.orc_factors_get.argtypes = [vp, C.c_int, ip, dp]
        L.orc_linearize.argtypes = [vp, vp, vp, C.POINTER(Setting), dp, vp, dp, dp, dp, C.POINTER(C.c_uint64)]
        L.orc_error.argtypes = [vp, vp, vp, C.POINTER(Setting), dp, vp, dp]
        L.orc_align.argtypes = [vp, vp, vp, C.POINTER(Setting), dp, C.POINTER(Result), dp, dp, C.c_int, C.POINTER(C.c_int), dp]
        L.orc_max_threads.restype = C.c_int
        _LIB = L
    return _LIB


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a, shape_last=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape_last is not None:
        a = a.reshape(-1, shape_last)
    return a


def default_setting(**kw):
    s = Setting()
    lib().orc_default_setting(C.byref(s))
    for k, v in kw.items():
        if not hasattr(s, k):
            raise AttributeError(k)
        if k == "restrict_mask":
            for i in range(6):
                s.restrict_mask[i] = float(v[i])
        else:
            setattr(s, k, v)
    return s


def fast_floor(x):
    return lib().orc_fast_floor(float(x))


def voxelgrid_sampling(points, leaf):
    """util/downsampling.hpp:23-78 (serial). points (N,3) -> (M,3) float64, ascending voxel key order."""
    p = _f64(points, 3)
    out = np.empty_like(p)
    n = lib().orc_voxelgrid_sampling(_dp(p), len(p), float(leaf), _dp(out)) if len(p) else 0
    return out[:n].copy()


class Cloud:
    """PointCloud (+ KdTree over it when tree=True)."""

    def __init__(self, points, normals=None, covs=None, tree=True):
        self._pts = _f64(points, 3)
        n = len(self._pts)
        self._nrm = _f64(normals, 3)
        self._cov = None if covs is None else _f64(np.asarray(covs, dtype=np.float64).reshape(n, 9), 9)
        self.h = lib().orc_cloud_create(_dp(self._pts), _dp(self._nrm), _dp(self._cov), n, 1 if tree else 0)
        self.n = n

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_cloud_destroy(self.h)
            self.h = None

    def __len__(self):
        return self.n

    def knn(self, queries, k, num_threads=1):
        q = _f64(queries, 3)
        idx = np.empty((len(q), k), dtype=np.int64)
        sqd = np.empty((len(q), k), dtype=np.float64)
        lib().orc_knn(self.h, _dp(q), len(q), k, idx.ctypes.data_as(C.POINTER(C.c_int64)), _dp(sqd), num_threads)
        return idx, sqd

    def estimate_normals_covariances(self, num_neighbors=20, num_threads=1):
        lib().orc_estimate_normals_covariances(self.h, num_neighbors, num_threads)

    def get(self):
        pts = np.empty((self.n, 3))
        nrm = np.empty((self.n, 3))
        cov = np.empty((self.n, 9))
        lib().orc_cloud_get(self.h, _dp(pts), _dp(nrm), _dp(cov))
        return pts, nrm, cov.reshape(self.n, 3, 3)


class VoxelMap:
    """GaussianVoxelMap: VoxelMap(cloud, leaf) = one insert() of a cloud with covariances (registration_helper.cpp:50-54);
    VoxelMap(None, leaf) = empty map for incremental use: insert(cloud, T) any number of times, set_lru(horizon, clear_cycle)
    (incremental_voxelmap.hpp:46,55-92)."""

    def __init__(self, cloud, leaf):
        if cloud is None:
            self.h = lib().orc_voxelmap_new(float(leaf))
        else:
            self.h = lib().orc_voxelmap_create(cloud.h, float(leaf))
        self.n = lib().orc_voxelmap_size(self.h)

    def insert(self, cloud, T=None):
        t = None if T is None else np.ascontiguousarray(np.asarray(T, dtype=np.float64).T).ravel()  # column-major 4x4
        lib().orc_voxelmap_insert(self.h, cloud.h, None if t is None else _dp(t))
        self.n = lib().orc_voxelmap_size(self.h)

    def set_lru(self, horizon=100, clear_cycle=10):
        lib().orc_voxelmap_set_lru(self.h, int(horizon), int(clear_cycle))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_voxelmap_destroy(self.h)
            self.h = None

    def __len__(self):
        return self.n

    def set_search_offsets(self, n):
        lib().orc_voxelmap_set_search_offsets(self.h, n)

    def get(self):
        coords = np.empty((self.n, 3), dtype=np.int32)
        means = np.empty((self.n, 3))
        covs = np.empty((self.n, 9))
        counts = np.empty(self.n, dtype=np.uint64)
        lib().orc_voxelmap_get(self.h, coords.ctypes.data_as(C.POINTER(C.c_int)), _dp(means), _dp(covs), counts.ctypes.data_as(C.POINTER(C.c_uint64)))
        return coords, means, covs.reshape(self.n, 3, 3), counts


class FlatMap(VoxelMap):
    """IncrementalVoxelMap<FlatContainerCov> (flat_container.hpp, incremental_voxelmap.hpp): voxels that keep up to
    max_num_points_in_cell of the inserted points (at least sqrt(min_sq_dist_in_cell) apart) with their covariances."""

    def __init__(self, leaf):
        self.h = lib().orc_flatmap_new(float(leaf))
        self.n = 0

    def set_setting(self, min_sq_dist_in_cell=0.01, max_num_points_in_cell=10):
        lib().orc_flatmap_set_setting(self.h, float(min_sq_dist_in_cell), int(max_num_points_in_cell))

    def get(self):
        """coords (V,3), counts (V,), points (P,3), covs (P,3,3) with the points of voxel 0 first, then voxel 1, ..."""
        total = lib().orc_flatmap_total_points(self.h)
        coords = np.empty((self.n, 3), dtype=np.int32)
        counts = np.empty(self.n, dtype=np.uint64)
        pts = np.empty((total, 3))
        covs = np.empty((total, 9))
        lib().orc_flatmap_get(self.h, coords.ctypes.data_as(C.POINTER(C.c_int)), counts.ctypes.data_as(C.POINTER(C.c_uint64)), _dp(pts), _dp(covs))
        return coords, counts, pts, covs.reshape(total, 3, 3)


class Factors:
    def __init__(self, n):
        self.h = lib().orc_factors_create(n)
        self.n = n

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_factors_destroy(self.h)
            self.h = None

    def get(self, is_voxelmap=False):
        ti = np.empty(self.n, dtype=np.int64)
        m = np.empty((self.n, 9))
        lib().orc_factors_get(self.h, int(is_voxelmap), ti.ctypes.data_as(C.POINTER(C.c_int64)), _dp(m))
        return ti, m.reshape(self.n, 3, 3)


def _T16(T):
    return np.ascontiguousarray(np.asarray(T, dtype=np.float64).reshape(4, 4).T).reshape(16)  # column-major


def _target_handles(target):
    if isinstance(target, VoxelMap):
        return None, target.h
    return target.h, None


def linearize(target, source, setting, T, factors):
    tc, tv = _target_handles(target)
    H = np.empty(36)
    b = np.empty(6)
    e = C.c_double()
    ninl = C.c_uint64()
    t16 = _T16(T)
    rc = lib().orc_linearize(tc, tv, source.h, C.byref(setting), _dp(t16), factors.h, _dp(H), _dp(b), C.byref(e), C.byref(ninl))
    assert rc == 0
    return H.reshape(6, 6), b, e.value, ninl.value


def error(target, source, setting, T, factors):
    tc, tv = _target_handles(target)
    e = C.c_double()
    t16 = _T16(T)
    rc = lib().orc_error(tc, tv, source.h, C.byref(setting), _dp(t16), factors.h, C.byref(e))
    assert rc == 0
    return e.value


class AlignResult:
    pass


def align(target, source, setting, init_T=None):
    """Registration<>::align (registration.hpp:33-43) with the LM/GN optimizer; returns an object mirroring RegistrationResult."""
    tc, tv = _target_handles(target)
    init_T = np.eye(4) if init_T is None else init_T
    res = Result()
    cap = 64
    te = np.zeros(cap)
    tn = np.zeros(cap)
    tl = C.c_int()
    el = C.c_double()
    t16 = _T16(init_T)
    rc = lib().orc_align(tc, tv, source.h, C.byref(setting), _dp(t16), C.byref(res), _dp(te), _dp(tn), cap, C.byref(tl), C.byref(el))
    assert rc == 0
    r = AlignResult()
    r.T_target_source = np.array(res.T).reshape(4, 4).T.copy()
    r.converged = bool(res.converged)
    r.iterations = int(res.iterations)
    r.num_inliers = int(res.num_inliers)
    r.H = np.array(res.H).reshape(6, 6)
    r.b = np.array(res.b)
    r.error = float(res.error)
    r.trace_e = te[: tl.value].copy()
    r.trace_new_e = tn[: tl.value].copy()
    r.elapsed_sec = el.value
    return r


def eigen_sym3(m, method=0):
    m = _f64(m).reshape(9)
    w = np.empty(3)
    v = np.empty(9)
    lib().orc_eigen_sym3(_dp(m), method, _dp(w), _dp(v))
    return w, v.reshape(3, 3)


def se3_exp(twist):
    a = _f64(twist).reshape(6)
    T = np.empty(16)
    lib().orc_se3_exp(_dp(a), _dp(T))
    return T.reshape(4, 4).T.copy()


def ldlt_solve(A, rhs):
    A = _f64(A).reshape(36)
    r = _f64(rhs).reshape(6)
    x = np.empty(6)
    lib().orc_ldlt_solve(_dp(A), _dp(r), _dp(x))
    return x


def max_threads():
    return lib().orc_max_threads()


def read_ply(path):
    """benchmark/read_points.hpp:52-109: binary little-endian PLY whose vertex properties are all float32; returns (N,3) float32 xyz."""
    with open(path, "rb") as f:
        nprops, n = 0, 0
        while True:
            line = f.readline().decode("ascii", "replace").strip()
            if line == "end_header" or line == "":
                break
            if line.startswith("element"):
                tok = line.split()
                assert tok[1] == "vertex"
                n = int(tok[2])
            elif line.startswith("property"):
                assert line.split()[1] == "float"
                nprops += 1
        buf = np.frombuffer(f.read(4 * nprops * n), dtype="<f4").reshape(n, nprops)
    return np.ascontiguousarray(buf[:, :3])
