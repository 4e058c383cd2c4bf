#!/usr/bin/env python3
"""CPU simulation of a wave-cooperative ("packet") exact nearest-neighbour traversal on the C3 workload: how many box tests and
leaf scans does a tile of 64 neighbouring queries need when the whole wave walks ONE path (a node is opened when ANY lane needs
it) instead of 64 divergent ones?  Design study for csrc/kd_packet.hpp; nothing here is on the product path.

  python scripts/sim_packet.py [n] [tiles]
"""
import sys
import os
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from small_gicp_amd import synthetic  # noqa: E402


def build_tree(pts):
    n = len(pts)
    D = 0
    while (n + (1 << D) - 1) >> D > 8:
        D += 1
    order = np.arange(n)
    thr = np.zeros(1 << D, np.float32)
    axis = np.zeros(1 << D, np.int32)
    lo = np.zeros((2 << D, 3), np.float32)
    hi = np.zeros((2 << D, 3), np.float32)
    P = pts
    for d in range(D + 1):
        k = np.arange((1 << d) + 1, dtype=np.uint64)
        bounds = ((k * np.uint64(n)) >> np.uint64(d)).astype(np.int64)
        starts = bounds[:-1]
        Q = P[order]
        blo = np.minimum.reduceat(Q, starts, axis=0)
        bhi = np.maximum.reduceat(Q, starts, axis=0)
        lo[(1 << d):(2 << d)] = blo
        hi[(1 << d):(2 << d)] = bhi
        if d == D:
            break
        ax = np.argmax(bhi - blo, axis=1)
        seg = np.repeat(np.arange(1 << d), np.diff(bounds))
        coord = Q[np.arange(n), ax[seg]]
        perm = np.lexsort((coord, seg))
        order = order[perm]
        coord = coord[perm]
        kk = np.arange(1 << d, dtype=np.uint64)
        mid = (((2 * kk + 1) * np.uint64(n)) >> np.uint64(d + 1)).astype(np.int64)
        thr[(1 << d):(2 << d)] = coord[np.minimum(mid, n - 1)]
        axis[(1 << d):(2 << d)] = ax
    return dict(n=n, D=D, pts=P[order], thr=thr, axis=axis, lo=lo, hi=hi)


def leaf_of(t, q):
    node = np.ones(len(q), np.int64)
    for d in range(t["D"]):
        a = t["axis"][node]
        qa = q[np.arange(len(q)), a]
        node = 2 * node + (qa - t["thr"][node] >= 0)
    return node - (1 << t["D"])


def morton30(q, lo, inv):
    c = np.clip(np.floor((q - lo) * inv), 0, 1023).astype(np.uint64)

    def spread(v):
        v = (v | (v << np.uint64(16))) & np.uint64(0x030000FF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x0300F00F)
        v = (v | (v << np.uint64(4))) & np.uint64(0x030C30C3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x09249249)
        return v

    return spread(c[:, 0]) | (spread(c[:, 1]) << np.uint64(1)) | (spread(c[:, 2]) << np.uint64(2))


def box_d2(lo, hi, q):
    d = np.maximum(np.maximum(lo - q, q - hi), 0)
    return (d * d).sum(-1)


def packet_search(t, q, bound2, seed_d2=None, group_levels=0, stats=None):
    """q: (64,3).  Returns nearest position per lane.  Counts box tests and leaf scans."""
    D = t["D"]
    n = t["n"]
    L = len(q)
    best = np.full(L, bound2, np.float32) if seed_d2 is None else np.minimum(seed_d2, bound2).astype(np.float32)
    bi = np.full(L, -1)
    GD = D - group_levels
    stack = [1]
    tests = scans = 0
    while stack:
        node = stack.pop()
        d = node.bit_length() - 1
        tests += 1
        lb = box_d2(t["lo"][node], t["hi"][node], q)
        if not (lb <= best).any():
            continue
        if d >= GD:
            k0 = (node - (1 << d)) << (D - d)
            k1 = k0 + (1 << (D - d))
            first = (k0 * n) >> D
            end = (k1 * n) >> D
            scans += (D - d == 0) and 1 or (1 << (D - d))
            c = t["pts"][first:end]
            dd = ((c[None, :, :] - q[:, None, :]) ** 2).sum(-1)
            j = dd.argmin(1)
            m = dd[np.arange(L), j]
            upd = m < best
            best = np.where(upd, m, best)
            bi = np.where(upd, first + j, bi)
            continue
        a = t["axis"][node]
        right = (q[:, a] - t["thr"][node] >= 0).sum()
        near = 2 * node + (1 if 2 * right > L else 0)
        stack.append(near ^ 1)
        stack.append(near)
    if stats is not None:
        stats["tests"] += tests
        stats["scans"] += scans
    return bi, best


def lane_need(t, q, d2final):
    """Lower bound of a single query's own walk: leaves whose box is within its final distance (vectorised over lanes via a
    packet walk with per-lane FINAL bounds and counting per lane)."""
    D = t["D"]
    L = len(q)
    cnt = np.zeros(L, np.int64)
    tests = np.zeros(L, np.int64)
    stack = [1]
    while stack:
        node = stack.pop()
        d = node.bit_length() - 1
        lb = box_d2(t["lo"][node], t["hi"][node], q)
        want = lb <= d2final
        tests += want if d == 0 else 0
        if not want.any():
            continue
        if d == D:
            cnt += want
            continue
        tests += 2 * want  # both children tested by the lanes that opened this node
        stack.append(2 * node)
        stack.append(2 * node + 1)
    return cnt, tests


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    ntiles = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    tgt = synthetic.scene(n, 1)
    T = synthetic.gt_transform()
    srcw = synthetic.scene(n, 2).astype(np.float64)
    Ti = np.linalg.inv(T)
    src = (srcw @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
    t = build_tree(tgt)
    print("tree: n=%d D=%d" % (t["n"], t["D"]))
    lo = tgt.min(0)
    inv = 1024.0 / (tgt.max(0) - lo).max()
    leaf = leaf_of(t, src)
    key = (leaf.astype(np.uint64) << np.uint64(30)) | morton30(src, lo, inv)
    src = src[np.argsort(key, kind="stable")]
    rng = np.random.default_rng(0)
    tiles = rng.choice(n // 64, ntiles, replace=False)
    bound2 = np.float32(1.05 * 1.05)

    def pose(frac):
        # interpolate identity -> T_gt (good enough a stand-in for the LM trajectory)
        from scipy.spatial.transform import Rotation as R
        rv = R.from_matrix(T[:3, :3]).as_rotvec() * frac
        M = np.eye(4)
        M[:3, :3] = R.from_rotvec(rv).as_matrix()
        M[:3, 3] = T[:3, 3] * frac
        return M

    prev_best = {}
    for frac, seeded in ((0.0, False), (0.7, True), (0.97, True), (1.0, True)):
        M = pose(frac)
        for gl in (0, 1, 2):
            st = dict(tests=0, scans=0)
            need = []
            ntest = []
            ext = []
            for ti in tiles:
                p = src[ti * 64:(ti + 1) * 64].astype(np.float64)
                q = (p @ M[:3, :3].T + M[:3, 3]).astype(np.float32)
                seed_d2 = None
                if seeded and ti in prev_best:
                    pj = prev_best[ti]
                    c = t["pts"][np.maximum(pj, 0)]
                    seed_d2 = np.where(pj >= 0, ((c - q) ** 2).sum(-1) * np.float32(1.0000002), np.inf).astype(np.float32)
                bi, best = packet_search(t, q, bound2, seed_d2, gl, st)
                if gl == 0:
                    c, tt = lane_need(t, q, best)
                    need.append(c.mean())
                    ntest.append(tt.mean())
                    ext.append(np.prod(np.sort(q.max(0) - q.min(0))[1:]))
                    prev_best[(frac, ti)] = bi
            if gl == 0:
                print("pose frac %.2f seeded=%d: per-lane need: leaves %.2f, box tests %.1f; tile area (2 largest extents) median %.2f m^2" % (frac, seeded, np.mean(need), np.mean(ntest), np.median(ext)))
            print("   group_levels=%d: packet box tests/tile %.1f, leaf scans/tile %.1f  -> VALU est %d (15/test, 80/leaf) | %d (15, 56)" %
                  (gl, st["tests"] / ntiles, st["scans"] / ntiles, (15 * st["tests"] + 80 * st["scans"]) / ntiles, (15 * st["tests"] + 56 * st["scans"]) / ntiles))
        for ti in tiles:
            prev_best[ti] = prev_best[(frac, ti)]


if __name__ == "__main__":
    main()
