#!/usr/bin/env python3
"""One C3 registration (10 LM iterations from the identity) on the bench's clouds: the unit bench.py re-runs under rocprofv3 to
measure the HBM traffic of K1.  Usage: one_registration.py [points]"""
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
pb = sga.Problem(sga.KdTree(tgt), src, np.eye(4))
st = sga.make_setting("GICP", max_correspondence_distance=1.0, max_iterations=10, rotation_eps=0.0, translation_eps=0.0)
r = pb.align(st, np.eye(4))
print("iterations", r.iterations, "inliers", r.num_inliers)
