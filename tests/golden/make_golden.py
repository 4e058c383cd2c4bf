"""Generates the committed fixtures under tests/golden/ (run in the build container, where /root/reference is mounted).

  c1_points.npz   raw xyz (float32) of the reference's data/target.ply and data/source.ply plus data/T_target_source.txt
                  (the GPU box has no /root/reference; the intensity channel is dropped, nothing else changes)
  c1_oracle.json  outputs of the CPU oracle (oracle/) on config C1 for every factor type: final pose, iterations,
                  num_inliers, final H/b/error and the per-iteration error trace — the 1e-4 parity anchor.

The oracle itself is pinned against the reference's own tolerances in tests/test_oracle_pins.py before these numbers are trusted.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402

REF = "/root/reference/data"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    tgt = orc.read_ply(os.path.join(REF, "target.ply"))
    src = orc.read_ply(os.path.join(REF, "source.ply"))
    T_gt = np.loadtxt(os.path.join(REF, "T_target_source.txt"))
    np.savez_compressed(os.path.join(OUT, "c1_points.npz"), target=tgt, source=src, T_target_source=T_gt)

    # config C1: RegistrationSetting defaults, downsampling 0.25 m, k = 10 (registration_helper.cpp:60-61), serial preprocessing
    td = orc.voxelgrid_sampling(tgt, 0.25)
    sd = orc.voxelgrid_sampling(src, 0.25)
    tc, sc = orc.Cloud(td), orc.Cloud(sd)
    tc.estimate_normals_covariances(10, 1)
    sc.estimate_normals_covariances(10, 1)
    out = {"downsampled_sizes": [len(td), len(sd)], "cases": {}}
    cases = [("GICP", orc.GICP, 0), ("PLANE_ICP", orc.PLANE_ICP, 0), ("ICP", orc.ICP, 0), ("HUBER_GICP", orc.GICP, 1), ("CAUCHY_GICP", orc.GICP, 2)]
    for name, kind, robust in cases:
        s = orc.default_setting(factor_kind=kind, robust_kind=robust, num_threads=1)
        r = orc.align(tc, sc, s)
        out["cases"][name] = dict(
            T=r.T_target_source.tolist(), converged=r.converged, iterations=r.iterations, num_inliers=r.num_inliers, H=r.H.tolist(), b=r.b.tolist(), error=r.error, trace_e=r.trace_e.tolist(), trace_new_e=r.trace_new_e.tolist()
        )
    vm = orc.VoxelMap(tc, 1.0)
    s = orc.default_setting(factor_kind=orc.GICP, num_threads=1)
    r = orc.align(vm, sc, s)
    out["cases"]["VGICP"] = dict(
        T=r.T_target_source.tolist(), converged=r.converged, iterations=r.iterations, num_inliers=r.num_inliers, H=r.H.tolist(), b=r.b.tolist(), error=r.error, trace_e=r.trace_e.tolist(), trace_new_e=r.trace_new_e.tolist(), num_voxels=len(vm)
    )
    with open(os.path.join(OUT, "c1_oracle.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", os.listdir(OUT))


if __name__ == "__main__":
    main()
