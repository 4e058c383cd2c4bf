#!/bin/bash
# same-box A/B of environment settings over the headline (C3), C2 and the odometry leg (C5): bash scripts/gpu_ab_all.sh "A=1" "A=0" ...
mkdir -p gpurun_out
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg timeout -s KILL 400 python bench.py --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --sustain-s 0 --no-fp64 --no-concurrent --no-vgicp --no-policy --no-traffic --no-preprocess --no-scaled --odom-frames ${ODOM:-40} > gpurun_out/aba_$i.json 2> gpurun_out/aba_$i.err
  tail -c 300 gpurun_out/aba_$i.err
  python - "$cfg" gpurun_out/aba_$i.json <<'PY'
import json, sys
j = json.load(open(sys.argv[2]))
r = j['roofline']; c = j['to_convergence']; p = j['plane_icp_c2']; o = j['kitti_odom']
print('%-22s C3 %.0f it/s (cold %.1f warm %.1f us; conv %.0f it/s) | C2 %.0f (cold %.1f warm %.1f) | C5 reg %.3f total %.3f pipe %.3f ms' % (sys.argv[1], j['value'], r['cold_pass_avg_us'], r['warm_pass_avg_us'], c['iterations_per_s'], p['value'], p['cold_pass_avg_us'], p['warm_pass_avg_us'], o['registration_ms_per_scan'], o['total_ms_per_scan'], o.get('pipelined_total_ms_per_scan', float('nan'))))
PY
done
