#!/usr/bin/env python3
"""CPU model: how many points of each warm pass of a C3 registration would have to search again if the certificate of a source point
stored C = 2 (as built) or C = 3 candidates ("a third stored candidate").  Needs no GPU.

A certificate of a point is {its C nearest target points, an exclusion radius rex}: every other target point is farther than rex.  After
the point has moved by m the nearest of the stored candidates is still the exact neighbour if its new distance is below rex - m (the
decision asks for pad * m more: certificate headroom).  A point that fails searches again and gets a new certificate.
The radius a search hands out is min(distance of the (C+1)-th nearest point, lower bound of everything the walk discarded); the second
term depends on the tree and is modelled from both sides:
  optimistic   rex = d_{C+1}                                   (the walk discarded nothing closer: fewest walkers)
  pessimistic  rex = min(d_{C+1}, d_1 + slack)                 (it discarded something right behind the explored ball: most walkers)
with slack = 0 for the cold passes and clamp(m, 3e-4, 0.02) for a re-walk (linearize.hip).  Poses: screw interpolation towards the ground
truth with the largest per-pass motions of the real run (profiles/r05_cert_pad.txt: 1.89, 0.90, 0.0725, 0.0105, 0.00173, 0.00029, 5e-5, 1e-5 m).
Usage: python scripts/sim_candidates.py [points] [pad]"""
import os
import sys
import time

import numpy as np
from scipy.spatial import cKDTree
from scipy.spatial.transform import Rotation

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from small_gicp_amd import synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
pad = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
WARM_DELTA = 0.1
target, source, T_gt = synthetic.registration_pair(n)
target, source = target.astype(np.float64), source.astype(np.float64)
t0 = time.time()
tree = cKDTree(target)
lo, hi = source.min(0), source.max(0)
corners = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])
rv = Rotation.from_matrix(T_gt[:3, :3]).as_rotvec()


def pose(f):
    T = np.eye(4)
    T[:3, :3] = Rotation.from_rotvec(rv * f).as_matrix()
    T[:3, 3] = T_gt[:3, 3] * f
    return T


def disp(A, B):
    return float(np.linalg.norm(corners @ (A[:3, :3] - B[:3, :3]).T + (A[:3, 3] - B[:3, 3]), axis=1).max())


total = disp(pose(1.0), pose(0.0))
steps = [1.89, 0.90, 0.0725, 0.0105, 0.00173, 0.00029, 5e-5, 1e-5, 0.0]
remaining = total - np.cumsum([0.0] + steps[:2])  # the first two steps as measured; afterwards the run converges onto ITS optimum:
fr = [0.0, steps[0] / total, (steps[0] + steps[1]) / total]
left = [0.0725 + 0.0105 + 0.00173 + 0.00029 + 5e-5 + 1e-5]
goal = fr[-1] + left[0] / total  # the pose the small steps converge to (a little short of / beyond the ground truth: irrelevant here)
acc = fr[-1]
for s in steps[2:]:
    acc += s / total
    fr.append(acc)
poses = [pose(f) for f in fr]
print("n = %d, pad = %.2f, kd-tree %.0f s; largest motion per pass:" % (n, pad, time.time() - t0), " ".join("%.5f" % disp(poses[k], poses[k - 1]) for k in range(1, len(poses))))

K = 5
results = {}
for C in (2, 3):
    for mode in ("optimistic", "pessimistic"):
        cand = np.full((n, C), -1, dtype=np.int64)
        rex = np.zeros(n)
        rows = []
        prev_q = None
        for k, T in enumerate(poses):
            q = source @ T[:3, :3].T + T[:3, 3]
            warm = k > 0 and disp(T, poses[k - 1]) <= WARM_DELTA
            if not warm:
                walk = np.ones(n, dtype=bool)
                moved = np.zeros(n)
            else:
                moved = np.linalg.norm(q - prev_q, axis=1)
                dc = np.stack([np.linalg.norm(target[cand[:, c]] - q, axis=1) for c in range(C)], axis=1)
                best = dc.min(axis=1)
                lim = rex - moved
                ok = best < lim - pad * moved
                walk = ~ok
                rex = np.where(ok, lim, rex)
            w = np.nonzero(walk)[0]
            if len(w):
                d, idx = tree.query(q[w], k=C + 1, workers=-1)
                cand[w] = idx[:, :C]
                r = d[:, C]
                if mode == "pessimistic":
                    slack = np.clip(moved[w], 3e-4, 0.02) if warm else 0.0
                    r = np.minimum(r, d[:, 0] + slack)
                rex[w] = r
            rows.append((warm, int(walk.sum())))
            prev_q = q
        results[(C, mode)] = rows
        print("C = %d, %-11s walkers per pass:" % (C, mode), " ".join(("%d" % w) if wm else "cold" for wm, w in rows), "(%.0f s)" % (time.time() - t0), flush=True)
print("measured on the MI355X, C = 2, pad 0.6 (profiles/r05_cert_pad.txt): cold cold cold 807812 58086 2091 89 6 1 | pad 0: cold cold cold 666181 61177 14021 2625 413 79")
