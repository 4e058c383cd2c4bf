#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
echo "=== suite"; timeout -s KILL 900 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -5
for cfg in "SGA_SPLIT_MIN_POINTS=8192" "SGA_SPLIT_MIN_POINTS=8192 SGA_SPLIT_PTS=4" "SGA_SPLIT_MIN_POINTS=131072"; do
  echo "=== $cfg"
  env $cfg timeout -s KILL 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --sustain-s 0 --no-fp64 --no-vgicp --no-traffic --no-policy --odom-frames 40 > gpurun_out/small.json 2> gpurun_out/small.err; tail -c 200 gpurun_out/small.err
  python - <<'PY'
import json
j=json.load(open('gpurun_out/small.json'))
c2=j['plane_icp_c2']; o=j['kitti_odom']
print('C3', round(j['value']), 'C2', round(c2['value']), 'k1', round(c2['k1_avg_us'],1), 'cold', round(c2['cold_pass_avg_us'],1), 'warm', round(c2['warm_pass_avg_us'],1), '| odom reg', round(o['registration_ms_per_scan'],3), 'total', round(o['total_ms_per_scan'],3), 'pipe', round(o.get('pipelined_total_ms_per_scan',0),3), 'cpp', o.get('cpp_driver',{}).get('registration_ms_per_scan'))
PY
done
