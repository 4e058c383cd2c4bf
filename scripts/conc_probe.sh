# GPU box: the concurrency-sensitive legs of the bench (side-by-side registrations, pipelined odometry) after every other leg has created
# and released its streams — the order in which the runtime's hardware queues are handed out is what this probes.
cd /root/repo
for cfg in "X=1" "GPU_MAX_HW_QUEUES=4" "X=2"; do
  env $cfg timeout -s KILL 600 python bench.py --no-preprocess --no-scaled --no-fp64 --sustain-s 0 --no-traffic --odom-frames 60 > gpurun_out/probe.json 2> gpurun_out/probe.err
  python -c "
import json; j=json.load(open('gpurun_out/probe.json')); c=j['concurrent_registrations']['jobs']; o=j['kitti_odom']; print('[$cfg]', {k: round(v['iterations_per_s']) for k,v in c.items()}, 'odom total %.3f reg %.3f pipelined %s' % (o['total_ms_per_scan'], o['registration_ms_per_scan'], {k: round(v['ms_per_scan'],3) for k,v in o['pipelined_by_workers'].items()}))"
done
