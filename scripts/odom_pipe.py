import sys, ctypes as C; sys.path.insert(0, "/root/repo")
from small_gicp_amd import odometry, api
for rep in range(3):
    p = odometry.run_synthetic_pipelined(48)
    st = (C.c_uint64 * 5)(); api.load().sga_allocator_stats(st)
    print("pipelined %.3f ms/scan; allocator malloc/stream/pool/pending/deferred" % p["ms_per_scan"], list(st))
