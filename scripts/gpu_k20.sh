#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
F="--no-cpu-baseline --no-policy --no-plane --no-vgicp --odom-frames 0 --no-traffic --no-fp64 --sustain-s 0"
for cfg in "W=5" "W=10" "W=20" "W=5 SGA_WARM_SPLIT=0" "W=5 SGA_GRID=0" "W=5"; do
  eval export $cfg
  env $cfg python bench.py --gpus 1 --steps 20 --warmup $W $F > gpurun_out/k20.json 2> gpurun_out/k20.err
  python -c "
import json; j=json.load(open('gpurun_out/k20.json')); print('$cfg', round(j['value']), j['steps'], round(j['ms_per_step'],4))"
  unset SGA_WARM_SPLIT SGA_GRID
done
