#!/usr/bin/env python3
"""One timed run of the CPU baseline in a process of its own, so that the OpenMP placement can be chosen per run (libgomp reads
OMP_PROC_BIND / OMP_PLACES once, when it is loaded): cpu_ref_rate.py <clouds.npz> <threads> <iterations> [tree_threads].
clouds.npz: tp, tcov, sp, scov (float64).  Prints one JSON line {"iterations_per_s": ...}."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import orc, ref  # noqa: E402

d = np.load(sys.argv[1])
threads, iters = int(sys.argv[2]), int(sys.argv[3])
tree_threads = int(sys.argv[4]) if len(sys.argv) > 4 else min(32, os.cpu_count() or 1)
if ref.available():
    a, b = ref.Cloud(d["tp"], None, d["tcov"], tree=True, tree_threads=tree_threads), ref.Cloud(d["sp"], None, d["scov"], tree=False)
    r = ref.align(a, b, ref.GICP, 1.0, 1.0, threads, iters, 0.0, 0.0)
else:
    orc.build()
    a, b = orc.Cloud(d["tp"], None, d["tcov"], tree=True), orc.Cloud(d["sp"], None, d["scov"], tree=False)
    r = orc.align(a, b, orc.default_setting(factor_kind=orc.GICP, num_threads=threads, max_iterations=iters, rotation_eps=0.0, translation_eps=0.0))
print(json.dumps({"iterations_per_s": (r.iterations + 1) / r.elapsed_sec, "threads": threads, "bind": os.environ.get("OMP_PROC_BIND"), "places": os.environ.get("OMP_PLACES")}))
