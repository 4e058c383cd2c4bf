#!/bin/bash
# what a 1-GPU box can run of the N-rank bench path: torchrun with ONE rank and --force-dist: NCCL process group, native RCCL communicator (sga_comm_init), ncclAllReduce per pass
mkdir -p gpurun_out
cd /root/repo
timeout -s KILL 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --force-dist --steps 200 --warmup 20 --no-cpu-baseline --no-fp64 --sustain-s 0 --odom-frames 12 --no-policy > gpurun_out/bench_force_dist.json 2> gpurun_out/bench_force_dist.err
tail -c 600 gpurun_out/bench_force_dist.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/bench_force_dist.json'))
print('value', j['value'], 'n_gpus', j['n_gpus'], 'config', json.dumps(j['config'])[:600])
print('per_rank', json.dumps(j.get('per_rank'))[:600])
print('odom', {k:v for k,v in j.get('kitti_odom',{}).items() if 'ms' in k})
PY
