#!/usr/bin/env python3
"""Aggregate rate of J independent C3 registrations running side by side on ONE GPU (one context = one stream and one host thread each; the
target index and the source cloud are shared, every job has its own factor state): how much of a lone registration's idle time — the
drain of its cold passes, the launch gaps, the row reduction, the host's 6x6 solve — other jobs can fill.  python scripts/concurrent_jobs.py [points] [regs]"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import small_gicp_amd as sga

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
regs = int(sys.argv[2]) if len(sys.argv) > 2 else 30
target, source, T_gt = sga.synthetic.registration_pair(n)
ctx0 = sga.default_context()
tgt, src = sga.PointCloud(target), sga.PointCloud(source)
sga.estimate_covariances(tgt, None, 20)
sga.estimate_covariances(src, None, 20)
tree = sga.KdTree(tgt)
st = sga.make_setting("GICP", max_correspondence_distance=1.0, max_iterations=10, rotation_eps=0.0, translation_eps=0.0)
ctx0.synchronize()
sys.setswitchinterval(2e-5)
import ctypes as C
from small_gicp_amd import _lib
def astats():
    b = (C.c_uint64 * 5)(); _lib.load().sga_allocator_stats(b); return list(b)  # malloc, stream hits, pool hits, pending hits, deferred
extra = [sga.Context(0) for _ in range(int(os.environ.get("EXTRA_CONTEXTS", "0")))]
for J in (1, 2, 3, 4):
    ctxs = [sga.Context(0) for _ in range(J)]
    pbs = [sga.Problem(tree, src, np.eye(4), ctx=c) for c in ctxs]
    poses = [None] * J
    for j in range(J):
        for _ in range(2):
            pbs[j].align(st, np.eye(4))
    start = threading.Barrier(J + 1)
    def work(j):
        start.wait()
        for _ in range(regs):
            poses[j] = pbs[j].align(st, np.eye(4)).T_target_source
        ctxs[j].synchronize()
    ths = [threading.Thread(target=work, args=(j,)) for j in range(J)]
    for t in ths: t.start()
    a0 = astats()
    start.wait()
    t0 = time.perf_counter()
    for t in ths: t.join()
    wall = time.perf_counter() - t0
    a1 = astats()
    print("   allocator during the run (malloc, stream hits, pool hits, pending hits, deferred):", [y - x for x, y in zip(a0, a1)])
    same = all(np.array_equal(poses[0], p) for p in poses)
    print("J=%d: %d registrations x 10 iterations in %.1f ms -> %.0f iterations/s aggregate (%.0f per job), poses identical across jobs: %s" % (J, J * regs, wall * 1e3, J * regs * 10 / wall, regs * 10 / wall, same), flush=True)
    del pbs, ctxs
