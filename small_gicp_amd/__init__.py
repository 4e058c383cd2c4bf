"""small_gicp_amd — MI355X-native drop-in for the per-iteration registration hot path of koide3/small_gicp.

The compute lives in lib/libsmall_gicp_amd.so (hand-written HIP for gfx950 behind the C-ABI of include/small_gicp_amd.h).
This package is the thin host layer: ctypes binding (_lib), a Python mirror of the reference's module (api) and the frozen
synthetic workloads of the benchmark configs (synthetic).
"""
from . import api, synthetic  # noqa: F401
from ._lib import GICP, ICP, LIB_PATH, PLANE_ICP, SgaError, load  # noqa: F401
from .api import (  # noqa: F401
    Context,
    GaussianVoxelMap,
    IncrementalVoxelMapCov,
    KdTree,
    MultiProblem,
    PointCloud,
    Problem,
    RegistrationResult,
    align,
    default_context,
    estimate_covariances,
    estimate_normals,
    estimate_normals_covariances,
    error_model_eval,
    get_warm_limit,
    set_error_model,
    set_grid_mode,
    set_search_mode,
    set_warm_limit,
    make_setting,
    optimize,
    pinned_copy,
    pinned_empty,
    preprocess_points,
    unpack_accumulator,
    voxelgrid_sampling,
)

__version__ = "0.1.0"
