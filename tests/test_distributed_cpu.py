"""The N > 1 path on CPU: world_size-2 `gloo` processes shard the source cloud, keep the target replicated, all-reduce the
30-double accumulator once per linearize and one double per error pass, and every rank runs the product's host optimizer
(sga_optimize) on the reduced numbers — exactly bench.py's multi-GPU structure with the oracle standing in for the kernels
(there is no GPU in this container).  The result must equal the single-process registration."""
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
sp, sn, scov = sc_full.get()
lo, hi = rank * len(sp) // world, (rank + 1) * len(sp) // world      # contiguous source shard, target replicated
shard = orc.Cloud(sp[lo:hi], sn[lo:hi], scov[lo:hi], tree=False)
s = orc.default_setting(factor_kind=orc.GICP, num_threads=1)
f = orc.Factors(len(shard))
def pack(H, b, e, n):
    acc = np.zeros(30); k = 0
    for i in range(6):
        for j in range(i, 6):
            acc[k] = H[i, j]; k += 1
    acc[21:27] = b; acc[27] = e; acc[28] = n
    return acc
def lin(T):
    acc = torch.from_numpy(pack(*orc.linearize(tc, shard, s, T, f)))
    dist.all_reduce(acc)
    return sga.unpack_accumulator(acc.numpy())
def err(T):
    e = torch.tensor([orc.error(tc, shard, s, T, f)], dtype=torch.float64)
    dist.all_reduce(e)
    return float(e[0])
res = sga.optimize(sga.make_setting("GICP"), np.eye(4), lin, err)
ts = torch.from_numpy(res.T_target_source.copy()); ref = ts.clone(); dist.broadcast(ref, 0)
assert torch.equal(ts, ref), "ranks diverged"
if rank == 0:
    print("RESULT " + json.dumps(dict(T=res.T_target_source.tolist(), iterations=res.iterations, num_inliers=res.num_inliers, error=res.error)))
dist.destroy_process_group()
"""


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
    assert abs(r["error"] - g["error"]) <= 1e-9 * abs(g["error"])
