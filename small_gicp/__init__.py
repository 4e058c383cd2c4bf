"""`import small_gicp` — the reference's Python module name and call signatures (src/python/*.cpp of koide3/small_gicp, pybind11)
served by the MI355X engine (small_gicp_amd, through the C-ABI of include/small_gicp_amd.h).

Scope: the subset exercised by the reference's own src/test/python_test.py and src/example/basic_registration.py
(SURVEY.md §8f row 1): PointCloud, KdTree, GaussianVoxelMap, read_ply, voxelgrid_sampling, estimate_normals /
estimate_covariances / estimate_normals_covariances, preprocess_points, the three align() overloads, RegistrationResult,
DistanceRejector and the per-point factors ICPFactor / PointToPlaneICPFactor / GICPFactor.

Argument names, order and defaults are the binding's (file:line cited per function).  `num_threads` is accepted and ignored:
the work runs on the GPU.
"""
import numpy as np

import small_gicp_amd.api as _api
from small_gicp_amd import io as _io
from small_gicp_amd.api import GaussianVoxelMap, IncrementalVoxelMapCov, KdTree, PointCloud, RegistrationResult  # noqa: F401

_DEG01 = 0.1 * np.pi / 180.0


def read_ply(filename):
    """misc.cpp:18-27: read a simple PLY file -> PointCloud."""
    return PointCloud(_io.read_ply(filename))


# ---- preprocessing (src/python/preprocess.cpp) ------------------------------------------------------------------------------------
def voxelgrid_sampling(points, downsampling_resolution, num_threads=1):
    """preprocess.cpp:24-34 (numpy Nx3 / Nx4) and :63-79 (PointCloud)."""
    return _api.voxelgrid_sampling(points if isinstance(points, PointCloud) else PointCloud(np.asarray(points)), downsampling_resolution)


def estimate_normals(points, tree=None, num_neighbors=20, num_threads=1):
    """preprocess.cpp:108-124."""
    _api.estimate_normals(points, tree, num_neighbors)


def estimate_covariances(points, tree=None, num_neighbors=20, num_threads=1):
    """preprocess.cpp:142-158."""
    _api.estimate_covariances(points, tree, num_neighbors)


def estimate_normals_covariances(points, tree=None, num_neighbors=20, num_threads=1):
    """preprocess.cpp:176-192."""
    _api.estimate_normals_covariances(points, tree, num_neighbors)


def preprocess_points(points, downsampling_resolution=0.25, num_neighbors=10, num_threads=1):
    """preprocess.cpp:210-236 (numpy) and :261-277 (PointCloud) -> (PointCloud, KdTree)."""
    return _api.preprocess_points(points, downsampling_resolution, num_neighbors)


# ---- registration (src/python/align.cpp) ----------------------------------------------------------------------------------------------
def _setting(registration_type, max_correspondence_distance, max_iterations, rotation_epsilon, translation_epsilon, verbose):
    return _api.make_setting(registration_type, max_correspondence_distance, max_iterations, rotation_eps=rotation_epsilon, translation_eps=translation_epsilon, verbose=verbose)


def _align_points(
    target_points,
    source_points,
    init_T_target_source=None,
    registration_type="GICP",
    voxel_resolution=1.0,
    downsampling_resolution=0.25,
    max_correspondence_distance=1.0,
    num_threads=1,
    max_iterations=20,
    rotation_epsilon=_DEG01,
    translation_epsilon=1e-3,
    verbose=False,
):
    """align.cpp:24-106 -> registration_helper.cpp:57-69: preprocess both clouds (k = 10), then register."""
    tgt, tree = _api.preprocess_points(np.asarray(target_points), downsampling_resolution, 10)
    src, _ = _api.preprocess_points(np.asarray(source_points), downsampling_resolution, 10)
    if registration_type == "VGICP":
        vm = GaussianVoxelMap(voxel_resolution, ctx=tgt.ctx)
        vm.insert(tgt)
        # registration_helper.cpp:125-137 leaves the rejector at its default 1.0 m^2 for VGICP
        s = _setting("GICP", 1.0, max_iterations, rotation_epsilon, translation_epsilon, verbose)
        return _api.Problem(vm, src, init_T_target_source).align(s, init_T_target_source)
    s = _setting(registration_type, max_correspondence_distance, max_iterations, rotation_epsilon, translation_epsilon, verbose)
    return _api.Problem(tree, src, init_T_target_source).align(s, init_T_target_source)


def _align_clouds(
    target,
    source,
    target_tree=None,
    init_T_target_source=None,
    registration_type="GICP",
    max_correspondence_distance=1.0,
    num_threads=1,
    max_iterations=20,
    rotation_epsilon=_DEG01,
    translation_epsilon=1e-3,
    verbose=False,
):
    """align.cpp:148-195 -> registration_helper.cpp:81-122: ICP / PLANE_ICP / GICP between preprocessed clouds."""
    if target_tree is None:
        target_tree = KdTree(target)
    target_tree.refresh_attributes()
    s = _setting(registration_type, max_correspondence_distance, max_iterations, rotation_epsilon, translation_epsilon, verbose)
    return _api.Problem(target_tree, source, init_T_target_source).align(s, init_T_target_source)


def _align_voxelmap(
    target_voxelmap,
    source,
    init_T_target_source=None,
    max_correspondence_distance=1.0,
    num_threads=1,
    max_iterations=20,
    rotation_epsilon=_DEG01,
    translation_epsilon=1e-3,
    verbose=False,
):
    """align.cpp:233-263: VGICP against a GaussianVoxelMap.  Like the binding (align.cpp:246) — and unlike the C++ helper of
    registration_helper.cpp:125-137 — this overload applies max_correspondence_distance to the rejector."""
    s = _setting("GICP", max_correspondence_distance, max_iterations, rotation_epsilon, translation_epsilon, verbose)
    return _api.Problem(target_voxelmap, source, init_T_target_source).align(s, init_T_target_source)


def align(*args, **kwargs):
    """The three overloads of small_gicp.align, selected by the type of the first argument like pybind11 does."""
    first = args[0] if args else kwargs.get("target_points", kwargs.get("target", kwargs.get("target_voxelmap")))
    if isinstance(first, GaussianVoxelMap):
        return _align_voxelmap(*args, **kwargs)
    if isinstance(first, PointCloud):
        return _align_clouds(*args, **kwargs)
    return _align_points(*args, **kwargs)


# ---- factors (src/python/factors.cpp) ---------------------------------------------------------------------------------------------------
class DistanceRejector:
    """factors.cpp:23-42, registration/rejector.hpp:19-28: reject iff sq_dist > max_dist_sq (default 1.0)."""

    def __init__(self):
        self.max_dist_sq = 1.0

    def set_max_distance(self, dist):
        self.max_dist_sq = float(dist) ** 2


class _PointFactor:
    """Per-point `linearize(target, source, kdtree, T, source_index, rejector) -> (success, H 6x6, b 6, e)` (factors.cpp:52-101).

    The binding evaluates one source point per call.  Here the first call for a given (target, source, tree, T, rejector) runs the
    ENGINE once over all source points — sga_linearize_per_point: the GPU search + the factor kernel's per-pair algebra, exported per
    point instead of summed — and later calls with other indices are served from that batch."""

    _kind = "ICP"

    def __init__(self):
        self._key = None
        self._batch = None

    def _evaluate(self, target, source, kdtree, T, rejector):
        kdtree.refresh_attributes()  # the tree may have been built before the covariances / normals were estimated
        s = _api.make_setting(self._kind)
        max_sq = getattr(rejector, "max_dist_sq", None)
        s.factor.max_dist_sq = -1.0 if max_sq is None or not np.isfinite(max_sq) else float(max_sq)
        return _api.Problem(kdtree, source, T).linearize_per_point(s.factor, T)

    def linearize(self, target, source, kdtree, T, source_index, rejector):
        key = (id(target), id(source), id(kdtree), np.asarray(T, dtype=np.float64).tobytes(), getattr(rejector, "max_dist_sq", None))
        if key != self._key:
            self._batch = self._evaluate(target, source, kdtree, T, rejector)
            self._key = key
        ok, H, b, e = self._batch
        i = int(source_index)
        if not ok[i]:
            return False, np.zeros((6, 6)), np.zeros(6), 0.0
        return True, H[i].copy(), b[i].copy(), float(e[i])


class ICPFactor(_PointFactor):
    _kind = "ICP"


class PointToPlaneICPFactor(_PointFactor):
    _kind = "PLANE_ICP"


class GICPFactor(_PointFactor):
    _kind = "GICP"


__all__ = [
    "PointCloud", "KdTree", "GaussianVoxelMap", "IncrementalVoxelMapCov", "RegistrationResult", "read_ply", "voxelgrid_sampling", "estimate_normals", "estimate_covariances",
    "estimate_normals_covariances", "preprocess_points", "align", "DistanceRejector", "ICPFactor", "PointToPlaneICPFactor", "GICPFactor",
]
