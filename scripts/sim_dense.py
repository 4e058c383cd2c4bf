"""CPU model of a dense (tile x candidate-block) cold search on the C3 clouds: how many candidates does a tile of 64
neighbouring queries need so that (nearly) all of them are settled exactly, when the candidates are whole kd groups (32 points = 4 leaves,
contiguous in kd order) taken nearest-first around the tile's bounding box?

A group g is a candidate of tile t when dist(box(g), bbox(t)) <= r_t; every target point outside the candidate set is then farther than
r_t from every query of the tile, so a query with d1 <= r_t (or r_t >= the search bound) is settled exactly.  Budget rule: the K nearest
groups; r_t = box distance of the first group left out.

  python scripts/sim_dense.py [n_points] [tile_stride]
"""
import sys
import time

import numpy as np
from scipy.spatial import cKDTree

sys.path.insert(0, ".")
from small_gicp_amd import synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 16
BOUND = 1.05  # search reach of the walks: rejector 1 m * (1 + kSearchMargin)

t0 = time.time()
target, source, T_gt = synthetic.registration_pair(n)
target = target.astype(np.float64)
source = source.astype(np.float64)

# ---- balanced implicit kd-tree like the engine's: node (d, k) owns [floor(k n / 2^d), floor((k+1) n / 2^d)), split at the median of the longest axis
D = max(0, int(np.ceil(np.log2(n / 8.0))))
order = np.arange(n)
thr = np.zeros(1 << D)
axis = np.zeros(1 << D, dtype=np.int8)
for d in range(D):
    for k in range(1 << d):
        lo, hi = (k * n) >> d, ((k + 1) * n) >> d
        idx = order[lo:hi]
        p = target[idx]
        ax = int(np.argmax(p.max(axis=0) - p.min(axis=0)))
        mid = (((2 * k + 1) * n) >> (d + 1)) - lo
        part = np.argpartition(p[:, ax], mid)
        order[lo:hi] = idx[part]
        node = (1 << d) + k
        thr[node] = p[part[mid], ax]
        axis[node] = ax
kd = target[order]
print("tree: depth %d, %.1f s" % (D, time.time() - t0))
G = 1 << (D - 2)  # groups: nodes of depth D - 2
gl = np.array([(g * n) >> (D - 2) for g in range(G + 1)])
glo = np.array([kd[gl[g]:gl[g + 1]].min(axis=0) for g in range(G)])
ghi = np.array([kd[gl[g]:gl[g + 1]].max(axis=0) for g in range(G)])
gcount = np.diff(gl)


def leaf_of(q):
    node = np.ones(len(q), dtype=np.int64)
    for d in range(D):
        qa = q[np.arange(len(q)), axis[node]]
        node = 2 * node + (qa >= thr[node])
    return node - (1 << D)


def morton10(q):
    lo, hi = target.min(axis=0), target.max(axis=0)
    c = np.clip(((q - lo) * (512.0 / (hi - lo).max())).astype(np.int64), 0, 1023)

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v

    return spread(c[:, 0]) | (spread(c[:, 1]) << 1) | (spread(c[:, 2]) << 2)


# the engine's source order: (target leaf at the initial pose, Morton code)
key = (leaf_of(source) << 30) | morton10(source)
sorder = np.argsort(key, kind="stable")
src = source[sorder]
tree = cKDTree(target)


def interp(f):
    from scipy.spatial.transform import Rotation

    T = np.eye(4)
    T[:3, :3] = Rotation.from_rotvec(Rotation.from_matrix(T_gt[:3, :3]).as_rotvec() * f).as_matrix()
    T[:3, 3] = T_gt[:3, 3] * f
    return T


tiles = np.arange(0, n // 64, stride)
for name, f in (("pass 1 (identity)", 0.0), ("pass 2", 0.55), ("pass 3", 0.9), ("at the optimum", 1.0)):
    T = interp(f)
    q = src @ T[:3, :3].T + T[:3, 3]
    d1, _ = tree.query(q, k=1, distance_upper_bound=np.inf)
    res = {K: dict(settled=0, total=0, tiles_open=0, cand=[], rt=[]) for K in (8, 12, 16, 24, 32)}
    need = []
    for t in tiles:
        qt = q[64 * t:64 * t + 64]
        dt = d1[64 * t:64 * t + 64]
        blo, bhi = qt.min(axis=0), qt.max(axis=0)
        gap = np.maximum(np.maximum(glo - bhi, blo - ghi), 0.0)
        gd = np.sqrt((gap * gap).sum(axis=1))
        near = np.argsort(gd)[:40]
        gds = gd[near]
        # groups needed to settle every query of the tile with the reach the rejector asks for
        rneed = min(BOUND, dt.max())
        need.append(int((gd <= rneed).sum()))
        for K, r in res.items():
            rt = min(gds[K], BOUND) if gds[K] < BOUND else BOUND
            kk = int((gds[:K] <= BOUND).sum())  # groups beyond the reach are never needed
            ok = (dt <= rt) | (rt >= BOUND)
            r["settled"] += int(ok.sum())
            r["total"] += 64
            r["tiles_open"] += int(not ok.all())
            r["cand"].append(int(gcount[near[:kk]].sum()))
            r["rt"].append(rt)
    need = np.array(need)
    print("%s: d1 mean %.3f p90 %.3f p99 %.3f max %.2f | groups a tile needs to settle all its queries: mean %.1f p50 %d p90 %d p99 %d max %d" % (
        name, d1.mean(), np.percentile(d1, 90), np.percentile(d1, 99), d1.max(), need.mean(), np.percentile(need, 50), np.percentile(need, 90), np.percentile(need, 99), need.max()))
    for K, r in res.items():
        print("   budget %2d groups: candidates mean %4.0f | queries settled %.4f | tiles with an open query %.3f | r_tile mean %.2f p10 %.2f" % (
            K, np.mean(r["cand"]), r["settled"] / r["total"], r["tiles_open"] / len(tiles), np.mean(r["rt"]), np.percentile(r["rt"], 10)))
print("total %.1f s" % (time.time() - t0))

# ---- where the candidates come from: groups touching the tile's own box, and within fixed margins (pass 1 and the optimum)
for name, f in (("pass 1", 0.0), ("optimum", 1.0)):
    T = interp(f)
    q = src @ T[:3, :3].T + T[:3, 3]
    ext, g0, g2, g5 = [], [], [], []
    for t in tiles:
        qt = q[64 * t:64 * t + 64]
        blo, bhi = qt.min(axis=0), qt.max(axis=0)
        gap = np.maximum(np.maximum(glo - bhi, blo - ghi), 0.0)
        gd = np.sqrt((gap * gap).sum(axis=1))
        ext.append(np.sort(bhi - blo))
        g0.append(int((gd <= 0).sum()))
        g2.append(int((gd <= 0.2).sum()))
        g5.append(int((gd <= 0.5).sum()))
    ext = np.array(ext)
    print("%s: tile box extents (sorted axes) median %s p90 %s | groups touching the box: median %d p90 %d; within 0.2 m: %d / %d; within 0.5 m: %d / %d" % (
        name, np.round(np.median(ext, axis=0), 2), np.round(np.percentile(ext, 90, axis=0), 2), np.median(g0), np.percentile(g0, 90), np.median(g2), np.percentile(g2, 90), np.median(g5), np.percentile(g5, 90)))
