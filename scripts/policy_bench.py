#!/usr/bin/env python3
"""The benchmark clouds through the reference's Registration<>::align with ParallelReductionHIP + HipAligned<LM> (oracle/_ref/policy_bench,
built where /root/reference is mounted) — bench.py's policy_c3 / policy_c2 legs call run().  Usage: policy_bench.py [GICP|PLANE_ICP|VGICP] [points] [reps] [num_gpus]"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, "oracle", "_ref", "policy_bench")


def run(kind="GICP", n=1_000_000, reps=5, num_gpus=1, clouds=None, timeout=600):
    """clouds: (target_xyz, source_xyz, target_attr, source_attr) float32 arrays, attr = n x 6 covariances (GICP) or n x 3 normals; None:
    the synthetic pair of that size with attributes estimated on the GPU (k = 20).  Returns the binary's JSON (None if it is not built)."""
    if not os.path.exists(BIN):
        return None
    import small_gicp_amd as sga

    if clouds is None:
        target, source, _ = sga.synthetic.registration_pair(n)
        tgt, src = sga.PointCloud(target), sga.PointCloud(source)
        if kind in ("GICP", "VGICP"):
            sga.estimate_covariances(tgt, None, 20)
            sga.estimate_covariances(src, None, 20)
            ta, sa = sga.api.sym6_from_mats(tgt.covs()), sga.api.sym6_from_mats(src.covs())
        else:
            sga.estimate_normals(tgt, None, 20)
            sga.estimate_normals(src, None, 20)
            ta, sa = tgt.normals()[:, :3], src.normals()[:, :3]
        clouds = (target, source, ta, sa)
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        paths = []
        for name, a in zip(("t", "s", "ta", "sa"), clouds):
            p = os.path.join(d, name + ".f32")
            np.ascontiguousarray(a, dtype="<f4").tofile(p)
            paths.append(p)
        env = dict(os.environ, OMP_NUM_THREADS="8")
        p = subprocess.run([BIN, kind, paths[0], paths[1], paths[2], paths[3], str(reps), str(num_gpus)], capture_output=True, text=True, timeout=timeout, env=env)
    for ln in p.stdout.splitlines():
        if ln.startswith("POLICY "):
            return json.loads(ln[7:])
    raise RuntimeError("policy_bench failed: rc %d\n%s\n%s" % (p.returncode, p.stdout[-2000:], p.stderr[-2000:]))


if __name__ == "__main__":
    kind = sys.argv[1] if len(sys.argv) > 1 else "GICP"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    g = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    print(json.dumps(run(kind, n, reps, g)))
