#!/bin/bash
# same-box A/B on the SMALL workloads: GICP at 100k <-> 100k (the C3 leg with --points 100000), C2 (point-to-plane 100k) and the odometry leg (C5)
mkdir -p gpurun_out
cd /root/repo
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg timeout -s KILL 300 python bench.py --points 100000 --steps ${STEPS:-400} --warmup 20 --no-cpu-baseline --sustain-s 0 --no-fp64 --no-vgicp --no-policy --no-traffic --odom-frames ${ODOM:-40} > gpurun_out/abs_$i.json 2> gpurun_out/abs_$i.err
  tail -c 300 gpurun_out/abs_$i.err
  python - "$cfg" gpurun_out/abs_$i.json <<'PY'
import json, sys
j = json.load(open(sys.argv[2]))
r = j['roofline']
print('%-28s GICP 100k %.0f it/s (cold %.1f warm %.1f us) | c2 %.0f (cold %.1f warm %.1f) | odom reg %.3f ms/scan' % (sys.argv[1], j['value'], r['cold_pass_avg_us'], r['warm_pass_avg_us'], j['plane_icp_c2']['value'], j['plane_icp_c2']['cold_pass_avg_us'], j['plane_icp_c2']['warm_pass_avg_us'], j['kitti_odom']['registration_ms_per_scan']))
PY
done
