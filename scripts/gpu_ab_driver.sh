#!/bin/bash
# same-box A/B of the DRIVER's bench line (--steps 20 --warmup 5: a 2.5 ms timed region right after half a registration of warm-up) under environment settings
mkdir -p gpurun_out
cd /root/repo
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg timeout -s KILL 200 python bench.py --gpus 1 --steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-cpu-baseline --sustain-s 0 --no-fp64 --no-vgicp --no-plane --no-policy --no-traffic --odom-frames 0 > gpurun_out/abd_$i.json 2> gpurun_out/abd_$i.err
  tail -c 200 gpurun_out/abd_$i.err
  python - "$cfg" gpurun_out/abd_$i.json <<'PY'
import json, sys
j = json.load(open(sys.argv[2]))
print('%-28s %.0f it/s  ms/step %.4f (timed region %.3f ms)' % (sys.argv[1], j['value'], j['ms_per_step'], j['ms_per_step'] * j['steps']))
PY
done
