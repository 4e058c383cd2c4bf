"""The reference's own Python test file and its Python example, UNCHANGED (tests/golden/reference_python_test/python_test.py =
src/test/python_test.py, basic_registration.py = src/example/basic_registration.py of koide3/small_gicp v1.0.1), run against this repository's `import small_gicp` module on the GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def write_ply(path, xyz):
    """Binary little-endian PLY with four float32 vertex properties, like the reference's data files (read_points.hpp:52-109)."""
    v = np.zeros((len(xyz), 4), "<f4")
    v[:, :3] = xyz
    with open(path, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nproperty float scalar_intensity\nend_header\n" % len(v)).encode())
        f.write(v.tobytes())


@pytest.mark.parametrize("name", ["python_test.py", "basic_registration.py"])
def test_reference_python_test_file_runs_unchanged(tmp_path, name):
    """python_test.py = src/test/python_test.py (the binding's test suite); basic_registration.py = src/example/basic_registration.py
    (the four usage examples of the README, each verified against the ground truth by the file's own pytest functions)."""
    d = np.load(os.path.join(GOLDEN, "c1_points.npz"))
    data = tmp_path / "data"
    data.mkdir()
    write_ply(data / "target.ply", d["target"])
    write_ply(data / "source.ply", d["source"])
    np.savetxt(data / "T_target_source.txt", d["T_target_source"])
    import filecmp
    import shutil

    fixture = os.path.join(GOLDEN, "reference_python_test", name)
    suite = tmp_path / name  # a byte-identical copy next to its data/ directory
    shutil.copyfile(fixture, suite)
    assert filecmp.cmp(fixture, suite, shallow=False)
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, "-m", "pytest", str(suite), "-q", "-x", "-p", "no:cacheprovider", "--rootdir", str(tmp_path)], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-4000:] + p.stderr[-2000:]
    assert " passed" in p.stdout and "failed" not in p.stdout, p.stdout[-1000:]
