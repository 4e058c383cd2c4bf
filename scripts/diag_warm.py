#!/usr/bin/env python3
"""Phases of the queue-fed warm kernel per pass (diagnostics build `make trips TRIPS_FLAGS=-DSGA_KD_NO_COUNT`,
SGA_LIB_PATH=small_gicp_amd/lib/libsmall_gicp_amd_trips.so): mean time a wave spends staging (certificate checks), walking and in
its factor stage, and when the waves start and end."""
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
pb = sga.Problem(tree, src)
lib = _lib.load()
buf = (C.c_ulonglong * 16)()
lib.sga_debug_kd_trips(buf)
k = [0]


def lin(T):
    r = pb.linearize(st.factor, T)
    lib.sga_debug_kd_trips(buf)
    waves = int(buf[15])
    if waves:
        wt = (C.c_ulonglong * (2 * min(waves, 32768)))()
        lib.sga_debug_kd_wave_times(wt, min(waves, 32768))
        w = np.array(wt, dtype=np.float64).reshape(-1, 2) * 0.01
        t0 = w[:, 0].min()
        print("pass %d: %d waves; per wave: staging %.1f us, walks %.1f us, factor stage %.1f us | starts: last %.1f us; ends: 50%% %.1f, 99%% %.1f, all %.1f us"
              % (k[0], waves, buf[12] * 0.01 / waves, buf[13] * 0.01 / waves, buf[14] * 0.01 / waves, (w[:, 0] - t0).max(), np.percentile(w[:, 1] - t0, 50), np.percentile(w[:, 1] - t0, 99), (w[:, 1] - t0).max()), flush=True)
    k[0] += 1
    return r


res = sga.optimize(st, np.eye(4), lin, lambda T: pb.error(st.factor, T))
lin(res.T_target_source)
lin(res.T_target_source)
