cd /root/repo; mkdir -p gpurun_out
for cfg in SGA_CHUNK_ADAPT=0 SGA_CHUNK_ADAPT=1; do
  env $cfg timeout -s KILL 400 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --sustain-s 0 --no-fp64 --no-traffic --odom-frames 100 > gpurun_out/ca_$cfg.json 2> gpurun_out/ca_$cfg.err
  python - $cfg gpurun_out/ca_$cfg.json <<'PY'
import json, sys
j = json.load(open(sys.argv[2]))
print(sys.argv[1], 'C3', round(j['value']), 'C2', json.dumps(j.get('plane_icp_c2'))[:330])
print('   C4', json.dumps(j.get('vgicp_c4'))[:200])
print('   C5', json.dumps(j.get('kitti_odom'))[:420])
PY
done
