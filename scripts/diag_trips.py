#!/usr/bin/env python3
"""Loop-body executions of the kd walk per linearization pass (diagnostics build: `make trips`, run with
SGA_LIB_PATH=small_gicp_amd/lib/libsmall_gicp_amd_trips.so SGA_SEARCH_QUEUE=0).  For every body: executions per wave of 64 queries,
and the share of lanes that were active in it."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import small_gicp_amd as sga
from small_gicp_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
target, source, T_gt = sga.synthetic.registration_pair(n)
tgt, src = sga.PointCloud(target), sga.PointCloud(source)
sga.estimate_covariances(tgt, None, 20)
sga.estimate_covariances(src, None, 20)
tree = sga.KdTree(tgt)
st = sga.make_setting("GICP", max_correspondence_distance=1.0, max_iterations=10, rotation_eps=0.0, translation_eps=0.0)
sga.set_search_mode(0)
pb = sga.Problem(tree, src)
lib = _lib.load()
lib.sga_debug_kd_trips.restype = C.c_int
names = ["uniform level", "pair step", "group header", "leaf scan", "pop iteration", "outer iteration", "leaf scan of the first group (SGA_KD_STALE builds)"]
k = [0]
buf = (C.c_ulonglong * 16)()
lib.sga_debug_kd_trips(buf)


def lin(T):
    r = pb.linearize(st.factor, T)
    lib.sga_debug_kd_trips(buf)
    waves = (n + 63) // 64
    print("pass %d: " % k[0] + "; ".join("%s %.1f/wave (%.0f%% lanes)" % (names[b], buf[8 + b] / waves, 100.0 * buf[b] / max(1, 64 * buf[8 + b])) for b in range(len(names)) if buf[8 + b]) + "; leaves per query %.2f" % (buf[3] / n), flush=True)
    wt = (C.c_ulonglong * (2 * min(waves, 32768)))()
    lib.sga_debug_kd_wave_times(wt, min(waves, 32768))
    w = np.array(wt, dtype=np.float64).reshape(-1, 2) * 0.01  # us (100 MHz)
    t0 = w[:, 0].min()
    start, end = w[:, 0] - t0, w[:, 1] - t0
    dur = end - start
    order = np.sort(end)
    print("        waves: duration mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f us | last start %.1f us, 50%% ended by %.1f, 90%% by %.1f, 99%% by %.1f, all by %.1f us"
          % (dur.mean(), np.percentile(dur, 50), np.percentile(dur, 90), np.percentile(dur, 99), dur.max(), start.max(), order[len(order) // 2], order[int(len(order) * 0.9)], order[int(len(order) * 0.99)], order[-1]), flush=True)
    k[0] += 1
    return r


sga.optimize(st, np.eye(4), lin, lambda T: pb.error(st.factor, T))
