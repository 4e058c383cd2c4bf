#!/bin/bash
# round-end record: the -m gpu suite, smoke(), the default bench (the driver's N=1 line), the driver's arguments, the profile
mkdir -p gpurun_out
cd /root/repo
echo "=== full suite"; timeout -s KILL 1200 python -m pytest tests/ -x -q -m gpu -rs 2>&1 | tail -6
echo "=== smoke"; timeout -s KILL 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2
echo "=== default bench"; ( time timeout -s KILL 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err ) 2>&1 | grep real; tail -c 300 gpurun_out/bench_default.err
echo "=== driver args"; timeout -s KILL 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_args.json 2> gpurun_out/bench_driver_args.err; tail -c 200 gpurun_out/bench_driver_args.err
python - <<'PY'
import json
for f in ('bench_default', 'bench_driver_args'):
    j=json.load(open('gpurun_out/%s.json' % f))
    r=j['roofline']
    print(f, 'value', round(j['value'],1), 'ms/step', round(j['ms_per_step'],4), 'frac', round(r['frac'],4), 'frac_dom', round(r['frac_dominant_kernel'],4), 'K1', round(r['avg_launch_us'],1), 'cold', round(r['cold_pass_avg_us'],1), 'warm', round(r['warm_pass_avg_us'],1), 'traffic', r['traffic'])
    print(' cpu', {k:v for k,v in j['cpu_baseline'].items() if k in ('value','cores')}, 'policy_c3', (j.get('policy_c3') or {}).get('inside_the_optimizer_iterations_per_s'), (j.get('policy_c3') or {}).get('policy_calls_iterations_per_s'))
    print(' c2', j['plane_icp_c2'].get('value'), 'c4', j['vgicp_c4'].get('value'), 'odom', {k:v for k,v in j['kitti_odom'].items() if 'ms_per_scan' in k}, 'conv', j['to_convergence'])
PY
echo "=== profile"; COMMIT=$(cat gpurun_out/.commit 2>/dev/null) bash scripts/profile_gpu.sh ${TAG:-r06} 2>&1 | tail -30
