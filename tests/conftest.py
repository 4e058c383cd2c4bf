import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
collect_ignore_glob = ["golden/*"]  # fixtures (incl. the reference's vendored python_test.py, run by test_reference_python_suite.py), not test modules


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with `pytest -m gpu`)")


def pose_error(T, T_ref):
    E = np.linalg.inv(T) @ T_ref
    dt = float(np.linalg.norm(E[:3, 3]))
    # the rotation angle from sine AND cosine: arccos alone resolves no angle below sqrt(2 eps) = 2.1e-8 rad (one ulp of the trace)
    R = E[:3, :3]
    s = 0.5 * float(np.linalg.norm([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]))
    dr = float(np.arctan2(s, (np.trace(R) - 1.0) / 2.0))
    return dt, dr


@pytest.fixture(scope="session")
def c1_raw():
    d = np.load(os.path.join(GOLDEN, "c1_points.npz"))
    return d["target"], d["source"], d["T_target_source"]


@pytest.fixture(scope="session")
def c1_gold():
    return json.load(open(os.path.join(GOLDEN, "c1_oracle.json")))


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as _orc

    _orc.build()
    return _orc


@pytest.fixture(scope="session")
def c1_oracle_clouds(orc, c1_raw):
    """Config C1 preprocessed by the oracle (serial path): 0.25 m voxel grid, normals + covariances k = 10."""
    tgt, src, _ = c1_raw
    td, sd = orc.voxelgrid_sampling(tgt, 0.25), orc.voxelgrid_sampling(src, 0.25)
    tc, sc = orc.Cloud(td), orc.Cloud(sd)
    tc.estimate_normals_covariances(10, 4)
    sc.estimate_normals_covariances(10, 4)
    return tc, sc


@pytest.fixture(scope="session")
def c1_f32(orc, c1_oracle_clouds):
    """The same clouds rounded to fp32 (what the GPU stores), plus oracle clouds built from exactly those fp32 values so that
    oracle-vs-GPU comparisons at a fixed pose isolate arithmetic from input rounding."""
    tc, sc = c1_oracle_clouds
    tp, tn, tcv = [a.astype(np.float32) for a in tc.get()]
    sp, sn, scv = [a.astype(np.float32) for a in sc.get()]
    otc = orc.Cloud(tp.astype(np.float64), tn.astype(np.float64), tcv.astype(np.float64))
    osc = orc.Cloud(sp.astype(np.float64), sn.astype(np.float64), scv.astype(np.float64), tree=False)
    return dict(tp=tp, tn=tn, tc=tcv, sp=sp, sn=sn, sc=scv, otc=otc, osc=osc)
