python - <<'PY'
import numpy as np, subprocess
d=np.load("tests/golden/c1_points.npz")
for n in ("target","source"): np.ascontiguousarray(d[n][:,:3],dtype="<f4").tofile("/tmp/%s.bin"%n)
p=subprocess.run(["oracle/_ref/test_reduction_hip","/tmp/target.bin","/tmp/source.bin"],capture_output=True,text=True)
print("\n".join(l for l in p.stdout.splitlines() if "VGICP" in l or "scan-to-model" in l or "FlatContainer" in l or "DONE" in l or "\"ok\": false" in l or "specialisation" in l)); print(p.stderr[-500:])
PY
