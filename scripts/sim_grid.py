#!/usr/bin/env python3
"""CPU model of the uniform cell grid on the C3 clouds (scipy): for cell edges 0.1 .. 0.25 m the candidates ring 1 scans per query (lane mean and
per-wave trip counts for queries sorted by cell), the share of queries it settles, and the rings / candidates the others need — the numbers
DESIGN.md section 3.9 quotes next to the measured counters.  Usage: python scripts/sim_grid.py"""
import sys, numpy as np, time
sys.path.insert(0, '/root/repo')
from small_gicp_amd import synthetic
from scipy.spatial import cKDTree
n = 1000000
tgt, src, T = synthetic.registration_pair(n)
tgt = tgt.astype(np.float64); src = src.astype(np.float64)
tree = cKDTree(tgt)
# pose "near": gt perturbed slightly like pass 3; "mid": like pass 2 (0.05 m, 0.3 deg off)
def pert(T, dt, ddeg):
    P = np.eye(4); P[:3,:3] = synthetic._rot([0.3,0.5,0.8], np.deg2rad(ddeg)); P[:3,3] = [dt*0.6, -dt*0.6, dt*0.5]
    return T @ P
for pose_name, Tq in (("gt", T), ("mid(0.05m,0.15deg)", pert(T, 0.05, 0.15))):
    q = src @ Tq[:3,:3].T + Tq[:3,3]
    d1, idx = tree.query(q, k=1, workers=8)
    print(pose_name, "NN dist quantiles", np.round(np.quantile(d1, [0.5, 0.9, 0.95, 0.99, 0.999]),3))
    for h in (0.1, 0.125, 0.15, 0.2, 0.25):
        lo = tgt.min(0) - h
        dims = np.ceil((tgt.max(0) - lo) / h).astype(int) + 2
        ct = np.floor((tgt - lo) / h).astype(int)
        key = (ct[:,2] * dims[1] + ct[:,1]) * dims[0] + ct[:,0]
        cnt = np.bincount(key, minlength=int(np.prod(dims)))
        cq = np.clip(np.floor((q - lo) / h).astype(int), 1, dims - 2)
        frac = (q - lo) / h - cq
        mf = np.minimum(frac, 1 - frac).min(1)
        minface = (1 + mf) * h
        cert = d1 < minface
        cs = np.concatenate([[0], np.cumsum(cnt)])
        base = (cq[:,2] * dims[1] + cq[:,1]) * dims[0] + cq[:,0]
        order = np.argsort(base, kind='stable')
        def trips(r, sel):
            # candidates in ring r for selected queries, wave-level trip counts after compaction of `sel`
            b = base[sel]; cx = cq[sel,0]
            rows = []
            for dz in range(-r, r+1):
                for dy in range(-r, r+1):
                    row = b + (dz * dims[1] + dy) * dims[0]
                    lo_i = np.clip(row - r, 0, len(cs)-1); hi_i = np.clip(row + r + 1, 0, len(cs)-1)
                    rows.append(cs[hi_i] - cs[lo_i])
            return np.array(rows)
        sel = order
        R1 = trips(1, sel)
        nw = len(sel)//64
        w1 = R1[:, :nw*64].reshape(R1.shape[0], nw, 64).max(2).sum(0)
        # stragglers: r* = ring needed
        un = ~cert
        rstar = np.ceil(np.maximum(d1 / h - mf, 1)).astype(int)   # ring with (r + mf) h >= d1
        rstar = np.where(d1 > 1.05, np.ceil(1.05/h).astype(int), rstar)
        s_sel = order[un[order]]
        rs = rstar[s_sel]
        # cost of stragglers: per compacted wave of 64: rows = (2 rmax+1)^2, candidates = sum over rows of wave-max
        tot_rows = 0; tot_trips = 0
        ns = len(s_sel)//64
        # approximate: each straggler scans its own ring r*; wave cost = max over lanes per row index — approximate by lane-sum max
        cand = np.zeros(len(s_sel))
        for r in np.unique(rs):
            m = rs == r
            if r > 12: continue
            cand[m] = trips(int(r), s_sel[m]).sum(0)
        cw = cand[:ns*64].reshape(ns, 64)
        rw = rs[:ns*64].reshape(ns, 64).max(1)
        print(f" h={h}: cells={np.prod(dims)/1e6:.0f}M cert1={cert.mean():.3f} ring1: lane-mean cand={R1.sum(0).mean():.0f} wave trips={w1.mean():.0f} | stragglers {un.mean()*100:.1f}%: r* mean={rs.mean():.2f} max={rs.max()} "
              f"cand lane-mean={cand.mean():.0f} wave-max-of-sums mean={cw.max(1).mean():.0f} rows/wave mean={((2*rw+1)**2).mean():.0f} -> est wave-instr: phase1={nw*(w1.mean()*16+200)/1e6:.1f}M stragglers={ns*(cw.max(1).mean()*16 + ((2*rw+1)**2).mean()*14)/1e6:.1f}M")
