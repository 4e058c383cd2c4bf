cd /tmp; export TMPDIR=/tmp
cat > /tmp/odom_run.py <<PY
import sys; sys.path.insert(0, "/root/repo")
from small_gicp_amd import odometry
r = odometry.run_synthetic(12)
PY
rm -rf /tmp/pv; timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -o odom -- python /tmp/odom_run.py > /tmp/pv.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/pv/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'vg_' in n or 'ds_' in n or 'downsample' in n or 'pack_cloud' in n: print("%-60s calls %s avg %.1f us" % (n[:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
