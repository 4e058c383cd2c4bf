#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
echo "=== default bench"; ( time timeout -s KILL 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err ) 2>&1 | grep real; tail -c 600 gpurun_out/bench_default.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/bench_default.json'))
r=j['roofline']
print('value', round(j['value'],1), 'ms/step', round(j['ms_per_step'],4), 'frac', round(r['frac'],4), 'frac_dom', round(r['frac_dominant_kernel'],4), 'K1', round(r['avg_launch_us'],1), 'launches', r['launches_timed'], 'cold', round(r['cold_pass_avg_us'],1), r['cold_passes_timed'], 'warm', round(r['warm_pass_avg_us'],1), r['warm_passes_timed'], 'traffic', r['traffic'])
print('cpu', {k:(v if not isinstance(v,str) else v[:80]) for k,v in j['cpu_baseline'].items() if k in ('value','cores','pinned_runs','by_thread_count','unpinned_median_at_best_thread_count')})
print('policy_c3', json.dumps(j.get('policy_c3'))[:900])
print('policy_c2', json.dumps(j.get('policy_c2'))[:600])
print('c2', j['plane_icp_c2'].get('value'), 'c4', j['vgicp_c4'].get('value'), 'odom', {k:v for k,v in j['kitti_odom'].items() if 'ms_per_scan' in k})
print('conv', j['to_convergence'])
PY
echo "=== 2 ranks on one device (callback transport)"; timeout -s KILL 600 python bench.py --gpus 2 --oversubscribe --steps 60 --warmup 10 --no-cpu-baseline --no-fp64 --sustain-s 0 --odom-frames 12 > gpurun_out/bench_2r.json 2> gpurun_out/bench_2r.err; tail -c 400 gpurun_out/bench_2r.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/bench_2r.json'))
print('value', j['value'], 'n_gpus', j['n_gpus'], 'per_rank', json.dumps(j.get('per_rank')), 'shard', j.get('sharded_vs_unsharded'))
print('odom', {k:v for k,v in j.get('kitti_odom',{}).items() if 'ms' in k}, 'pairs', j.get('kitti_odom_frame_pairs_per_rank'))
PY
python - <<'PY'
import json
j=json.load(open('gpurun_out/bench_default.json'))
print('helper_c1', j.get('helper_c1'), 'helper_c5', j.get('helper_c5'))
print('policy_c4', json.dumps(j.get('policy_c4'))[:500])
j=json.load(open('gpurun_out/bench_2r.json'))
print('scaling_model', json.dumps(j.get('scaling_model'))[:700])
PY
