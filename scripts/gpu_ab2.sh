#!/bin/bash
# same-box A/B of library builds / environment settings: C3 headline, C2 and the odometry leg.  bash scripts/gpu_ab2.sh "A=1" "SGA_LIB_PATH=..." ...
mkdir -p gpurun_out
cd /root/repo
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg timeout -s KILL 300 python bench.py --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --sustain-s 0 --no-fp64 --no-vgicp --no-policy --no-traffic --odom-frames ${ODOM:-40} > gpurun_out/ab_$i.json 2> gpurun_out/ab_$i.err
  tail -c 300 gpurun_out/ab_$i.err
  python - "$cfg" gpurun_out/ab_$i.json <<'PY'
import json, sys
j = json.load(open(sys.argv[2]))
r = j['roofline']
print('%-46s it/s %.0f  K1 %.1f us  cold %.1f  warm %.1f  | c2 %.0f | odom reg %.3f ms/scan' % (sys.argv[1].replace('/root/repo/small_gicp_amd/lib/', ''), j['value'], r['avg_launch_us'], r['cold_pass_avg_us'], r['warm_pass_avg_us'], j['plane_icp_c2']['value'], j['kitti_odom']['registration_ms_per_scan']))
PY
done
