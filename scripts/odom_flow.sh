# GPU box: the flow pipeline of C5 over (preprocessing workers) x (registration workers), one process per setting
cd /root/repo
for q in 8 16; do
for g in 1x1 2x1 2x2 3x2 3x3 4x3 4x4 5x4 6x4 6x6; do
  echo "== GPU_MAX_HW_QUEUES=$q $g"
  GPU_MAX_HW_QUEUES=$q timeout -s KILL 120 python scripts/odom_flow.py ${FRAMES:-60} ${PIN:-pageable} $g 2>&1 | grep -v "^sequential" 
done; done
