"""C5 through the C++ flow driver (examples/odometry_benchmark_flow.cpp) over (preprocessing workers) x (registration workers); poses against
the sequential C++ driver's.  usage: python scripts/odom_flow_cpp.py [frames] [PxR ...]   env: PIN=1 pinned scans, GPU_MAX_HW_QUEUES"""
import os
import sys
import tempfile

sys.path.insert(0, "/root/repo")
import numpy as np

from small_gicp_amd import odometry

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 100
grid = [tuple(int(x) for x in a.split("x")) for a in sys.argv[2:]] or [(1, 1), (2, 1), (2, 2), (3, 2), (3, 3), (4, 3), (4, 4), (6, 4), (8, 6)]
pinned = os.environ.get("PIN", "0") == "1"
work = tempfile.mkdtemp(prefix="sga_flow_")
seq = odometry.run_synthetic_cpp(frames, workdir=work)
print("sequential C++ driver: registration %.3f total %.3f ms/scan" % (seq["registration_ms_per_scan"], seq["total_ms_per_scan"]), flush=True)
for P, R in grid:
    r = odometry.run_synthetic_cpp_flow(frames, workdir=work, preprocess_workers=P, registration_workers=R, pinned=pinned, repeat=4)
    d = max(float(np.abs(a - b).max()) for a, b in zip(r["estimated"], seq["estimated"]))
    print("preprocess %d x registration %d: %.3f ms/scan (runs %s) latency %.2f ms, iterations %.2f, max |pose - sequential| %.1e" % (P, R, r["ms_per_scan"], " ".join("%.3f" % x for x in r["runs_ms_per_scan"]), r["frame_latency_ms"], r["mean_iterations"], d), flush=True)
