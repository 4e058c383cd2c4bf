"""C5 end to end, unprofiled: sequential and pipelined drivers, scans in pageable or pinned host memory.  Usage: odom_modes.py [frames]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402

from small_gicp_amd import odometry  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
out = {}
for pinned in (False, True):
    r = odometry.run_synthetic(frames, pinned=pinned)
    tag = "pinned" if pinned else "pageable"
    out["sequential_" + tag] = {k: r[k] for k in ("registration_ms_per_scan", "total_ms_per_scan", "mean_iterations")}
    ref = r["estimated"]
    for workers in (1, 2, 3):
        p = odometry.run_synthetic_pipelined(frames, pinned=pinned, workers=workers)
        same = all(np.array_equal(a, b) for a, b in zip(p["estimated"], ref))
        out["pipelined_%s_w%d" % (tag, workers)] = {"ms_per_scan": p["ms_per_scan"], "poses_identical": bool(same)}
print(json.dumps(out, indent=1))
