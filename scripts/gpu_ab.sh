#!/bin/bash
# GPU box: the -m gpu suite, then the headline bench under the environment settings given as arguments ("A=1 B=2" strings, one run each).
#   bash scripts/gpu_ab.sh "SGA_FAST_SCAN=1" "SGA_FAST_SCAN=0"
mkdir -p gpurun_out
cd /root/repo
if [ "${SKIP_TESTS:-0}" != "1" ]; then timeout -s KILL 900 python -m pytest tests/ -x -q -m gpu -rs 2>&1 | tail -15; fi
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg timeout -s KILL 300 python bench.py --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --sustain-s 0 --no-fp64 --no-concurrent --no-vgicp --odom-frames 0 > gpurun_out/ab_$i.json 2> gpurun_out/ab_$i.err
  tail -c 300 gpurun_out/ab_$i.err
  python - "$cfg" gpurun_out/ab_$i.json <<'PY'
import json, sys
j = json.load(open(sys.argv[2]))
r = j['roofline']
print('%-40s it/s %.0f  ms/step %.4f  K1 %.1f us  cold %.1f  warm %.1f (search %.1f)  frac %.3f  pose err %.2e' % (sys.argv[1], j['value'], j['ms_per_step'], r['avg_launch_us'], r['cold_pass_avg_us'], r['warm_pass_avg_us'], r['warm_pass_search_avg_us'], r['frac'], j['final_pose_error']['trans_m']))
PY
done
