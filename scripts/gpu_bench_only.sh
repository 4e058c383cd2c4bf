#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
( time timeout -s KILL 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err ) 2>&1 | grep real; tail -c 200 gpurun_out/bench_default.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/bench_default.json'))
r=j['roofline']
print('value', round(j['value'],1), 'ms/step', round(j['ms_per_step'],4), 'frac', round(r['frac'],4), 'K1', round(r['avg_launch_us'],1), 'cold', round(r['cold_pass_avg_us'],1), 'warm', round(r['warm_pass_avg_us'],1), 'traffic', r['traffic'])
for k in ('policy_c3','policy_c2','policy_c4'):
    p=j.get(k,{}); print(k, p.get('first_align_s'), p.get('whole_align_iterations_per_s'), p.get('policy_calls_iterations_per_s'), (p.get('lean') or {}).get('whole_align_iterations_per_s'))
print('c2', j['plane_icp_c2'].get('value'), 'c4', j['vgicp_c4'].get('value'), 'cpu', j['cpu_baseline']['value'], 'odom', j['kitti_odom']['registration_ms_per_scan'])
PY
