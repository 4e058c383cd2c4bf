#!/bin/bash
cd /root/repo
python - <<'PY'
import numpy as np, subprocess
d=np.load("tests/golden/c1_points.npz")
for n in ("target","source"): np.ascontiguousarray(d[n][:,:3],dtype="<f4").tofile("/tmp/%s.bin"%n)
for rep in range(25):
    p=subprocess.run(["oracle/_ref/test_reduction_hip","/tmp/target.bin","/tmp/source.bin"],capture_output=True,text=True)
    bad=[l for l in p.stdout.splitlines() if '"ok": false' in l]
    print(rep, p.returncode, bad[:3], p.stderr[-300:])
PY
