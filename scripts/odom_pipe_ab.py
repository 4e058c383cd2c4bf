"""pipelined C5 under different GIL switch intervals / worker counts (unprofiled)"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from small_gicp_amd import odometry
frames = 60
for sw in ("5e-3", "2e-5", "1e-4"):
    os.environ["SGA_PIPE_SWITCH_S"] = sw
    row = []
    for w in (1, 2, 3):
        for rep in range(2):
            r = odometry.run_synthetic_pipelined(frames, workers=w)
            row.append("w%d %.3f" % (w, r["ms_per_scan"]))
    print("switch interval %s s: %s" % (sw, "  ".join(row)), flush=True)
