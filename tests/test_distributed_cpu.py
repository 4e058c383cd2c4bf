"""The N > 1 path on CPU: world_size-2 `gloo` processes shard the source cloud, keep the target replicated and run the product's
protocol (bench.py --gpus N, csrc/comm.hip) with the oracle standing in for the kernels — there is no GPU in this container:
ONE all-reduce of 96 doubles per linearization (the system + the moments of the quadratic error model), no collective in the error
passes: every rank evaluates the trial errors from the reduced moments with the library's own host routine (sga_error_model_eval),
and every rank runs the product's host optimizer (sga_optimize) on the reduced numbers.  The result must equal the single-process
registration.  (The loop being partitioned: registration/reduction_omp.hpp:32-58.)"""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT

WORKER = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["SGA_ROOT"])
import torch, torch.distributed as dist
import small_gicp_amd as sga
from oracle import orc
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
d = np.load(os.path.join(os.environ["SGA_ROOT"], "tests", "golden", "c1_points.npz"))
td, sd = orc.voxelgrid_sampling(d["target"], 0.25), orc.voxelgrid_sampling(d["source"], 0.25)
tc, sc_full = orc.Cloud(td), orc.Cloud(sd)
tc.estimate_normals_covariances(10, 1); sc_full.estimate_normals_covariances(10, 1)
tp = tc.get()[0]
sp, sn, scov = sc_full.get()
lo, hi = rank * len(sp) // world, (rank + 1) * len(sp) // world      # contiguous source shard, target replicated
shard = orc.Cloud(sp[lo:hi], sn[lo:hi], scov[lo:hi], tree=False)
s = orc.default_setting(factor_kind=orc.GICP, num_threads=1)
f = orc.Factors(len(shard))
sys.path.insert(0, os.path.join(os.environ["SGA_ROOT"], "tests"))
from moments import accumulator96
state = {}
def lin(T):
    H, b, e, n = orc.linearize(tc, shard, s, T, f)
    ti, maha = f.get()
    acc = torch.from_numpy(accumulator96(H, b, e, n, T, sp[lo:hi, :3], tp[:, :3], ti, maha))
    dist.all_reduce(acc)                                              # the ONE collective of an LM iteration
    state["acc"], state["T"] = acc.numpy().copy(), T.copy()
    return sga.unpack_accumulator(state["acc"][:30])
def err(T):
    return sga.error_model_eval(state["acc"], state["T"], T)          # no collective: every rank holds the reduced moments
res = sga.optimize(sga.make_setting("GICP"), np.eye(4), lin, err)
ts = torch.from_numpy(res.T_target_source.copy()); ref = ts.clone(); dist.broadcast(ref, 0)
assert torch.equal(ts, ref), "ranks diverged"
if rank == 0:
    print("RESULT " + json.dumps(dict(T=res.T_target_source.tolist(), iterations=res.iterations, num_inliers=res.num_inliers, error=res.error)))
dist.destroy_process_group()
"""


def test_error_model_from_moments_equals_the_error_pass(orc, c1_oracle_clouds):
    """sga_error_model_eval on the 96-double accumulator == the oracle's error pass with the frozen correspondences (gicp_factor.hpp:80-89)."""
    import small_gicp_amd as sga
    from moments import accumulator96

    tc, sc = c1_oracle_clouds
    s = orc.default_setting(factor_kind=orc.GICP, num_threads=1)
    f = orc.Factors(len(sc))
    rng = np.random.default_rng(3)
    T = np.eye(4)
    T[:3, 3] = [0.05, -0.02, 0.01]
    H, b, e, n = orc.linearize(tc, sc, s, T, f)
    ti, maha = f.get()
    acc = accumulator96(H, b, e, n, T, sc.get()[0][:, :3], tc.get()[0][:, :3], ti, maha)
    assert abs(sga.error_model_eval(acc, T, T) - e) <= 1e-12 * abs(e)
    for _ in range(4):
        Tn = T @ orc.se3_exp(rng.normal(0, 0.02, 6))
        want = orc.error(tc, sc, s, Tn, f)
        got = sga.error_model_eval(acc, T, Tn)
        assert abs(got - want) <= 1e-9 * abs(want), (got, want)


def test_sharded_source_allreduce_gloo(tmp_path, c1_gold):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, SGA_ROOT=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29731", str(script)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][0]
    import json

    r = json.loads(line[len("RESULT "):])
    g = c1_gold["cases"]["GICP"]
    assert r["iterations"] == g["iterations"] and r["num_inliers"] == g["num_inliers"]
    assert np.allclose(np.array(r["T"]), np.array(g["T"]), atol=1e-9)
    assert abs(r["error"] - g["error"]) <= 1e-8 * abs(g["error"])
