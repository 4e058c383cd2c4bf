#!/usr/bin/env python3
"""Would starting the heavy tiles first shorten the search kernel's tail?  Per-tile work (max leaves over the 64 lanes) of every
pass, its correlation with the previous pass, and a list-scheduling simulation (8192 wave slots, duration = a + b * work) of the
natural order against heavy-first orders built from the previous pass.  Usage: python scripts/diag_lpt.py [points]"""
import heapq
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import small_gicp_amd as sga

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
target, source, T_gt = sga.synthetic.registration_pair(n)
tgt, src = sga.PointCloud(target), sga.PointCloud(source)
sga.estimate_covariances(tgt, None, 20)
sga.estimate_covariances(src, None, 20)
tree = sga.KdTree(tgt)
st = sga.make_setting("GICP", max_correspondence_distance=1.0, max_iterations=10, rotation_eps=0.0, translation_eps=0.0)
sga.set_search_mode(0)
pb = sga.Problem(tree, src)
pb.search_stats(True)
works = []
lanes = []


def lin(T):
    r = pb.linearize(st.factor, T)
    lv = pb.search_stats()
    m = len(lv) // 64 * 64
    works.append(lv[:m].reshape(-1, 64).max(axis=1).astype(np.float64))
    lanes.append(lv[:m].reshape(-1, 64).astype(np.float64))
    return r


sga.optimize(st, np.eye(4), lin, lambda T: pb.error(st.factor, T))

def makespan(dur, order, slots=8192):
    h = [0.0] * slots
    heapq.heapify(h)
    end = 0.0
    for t in order:
        s = heapq.heappop(h)
        e = s + dur[t]
        end = max(end, e)
        heapq.heappush(h, e)
    return end


for k in range(1, min(5, len(works))):
    w, prev = works[k], works[k - 1]
    dur = 10.0 + 6.0 * w  # us: ~55 us for the average wave
    nat = np.arange(len(w))
    thr = np.percentile(prev, 70)
    two = np.concatenate([nat[prev >= thr], nat[prev < thr]])
    full = np.argsort(-prev, kind="stable")
    oracle = np.argsort(-w, kind="stable")
    print("pass %d: corr(work, prev work) = %.2f | ideal (sum/slots) %.0f us | makespan natural %.0f, heavy-first two buckets (prev) %.0f, sorted by prev %.0f, sorted by own work (oracle) %.0f"
          % (k, np.corrcoef(w, prev)[0, 1], dur.sum() / 8192, makespan(dur, nat), makespan(dur, two), makespan(dur, full), makespan(dur, oracle)), flush=True)


# late tiles split into narrower waves (the drain phase has idle slots and idle VALUs): last `frac` of the tiles as 64 / parts lanes
for k in range(0, min(4, len(works))):
    L = lanes[k]
    base = makespan(10.0 + 6.0 * L.max(axis=1), np.arange(len(L)))
    out = []
    for frac in (0.25, 0.5):
        for parts in (2, 4):
            cut = int(len(L) * (1 - frac))
            d = list(10.0 + 6.0 * L[:cut].max(axis=1))
            sub = L[cut:].reshape(-1, parts, 64 // parts).max(axis=2).reshape(-1)
            d += list(10.0 + 6.0 * sub)
            d = np.array(d)
            out.append("last %.0f%% in %d parts: %.0f" % (100 * frac, parts, makespan(d, np.arange(len(d)))))
    print("pass %d: natural %.0f us | %s" % (k, base, " | ".join(out)), flush=True)

# static (pose-invariant) per-tile features of the sorted source: extent of the 64 points, and the same over the kd-ordered target
sp = pb.sorted_points()[:, :3]
m = len(sp) // 64 * 64
tiles = sp[:m].reshape(-1, 64, 3).astype(np.float64)
extent = np.linalg.norm(tiles.max(axis=1) - tiles.min(axis=1), axis=1)
print("static feature: tile extent (m) p50 %.2f p90 %.2f p99 %.2f max %.2f" % tuple(np.percentile(extent, [50, 90, 99, 100])))
for k in range(0, min(4, len(works))):
    w = works[k]
    dur = 10.0 + 6.0 * w
    order = np.argsort(-extent, kind="stable")
    print("pass %d: corr(work, extent) = %.2f, rank corr = %.2f | makespan natural %.0f, largest extent first %.0f, own work first (oracle) %.0f"
          % (k, np.corrcoef(w, extent)[0, 1], np.corrcoef(np.argsort(np.argsort(w)), np.argsort(np.argsort(extent)))[0, 1], makespan(dur, np.arange(len(w))), makespan(dur, order),
             makespan(dur, np.argsort(-w, kind="stable"))), flush=True)


