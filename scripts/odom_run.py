#!/usr/bin/env python3
"""The C5 odometry leg alone (12 synthetic KITTI-shaped scans) - the workload of the odometry kernel traces."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from small_gicp_amd import odometry

r = odometry.run_synthetic(int(sys.argv[1]) if len(sys.argv) > 1 else 12)
print({k: v for k, v in r.items() if k not in ("estimated", "ground_truth")})
