cd /root/repo
python - <<'PY'
import numpy as np, os
d = np.load("tests/golden/c1_points.npz")
for name in ("target", "source"):
    np.ascontiguousarray(d[name][:, :3], dtype="<f4").tofile("/tmp/%s.bin" % name)
PY
for i in 1 2 3 4; do oracle/_ref/test_reduction_hip /tmp/target.bin /tmp/source.bin | grep "moved by"; done
