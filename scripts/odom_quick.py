import sys; sys.path.insert(0, "/root/repo")
from small_gicp_amd import odometry
r = odometry.run_synthetic(14)
print("seq reg %.3f total %.3f iters %.2f rpe %.5f" % (r["registration_ms_per_scan"], r["total_ms_per_scan"], r["mean_iterations"], r["rpe_trans_m_mean"]))
