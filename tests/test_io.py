"""On-disk formats (small_gicp_amd/io.py) against the reference's own files and rules — CPU only."""
import os

import numpy as np

from conftest import ROOT
from small_gicp_amd import io


def test_bin_and_ply_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    p = rng.normal(0, 10, (1000, 3)).astype(np.float32)
    io.write_points(tmp_path / "a.bin", np.concatenate([p, rng.uniform(0, 1, (1000, 1)).astype(np.float32)], 1))
    q = io.read_points(tmp_path / "a.bin")
    assert q.shape == (1000, 4) and (q[:, :3] == p).all() and (q[:, 3] == 1).all()  # intensity replaced by w = 1 (read_points.hpp:28-30)
    io.write_ply(tmp_path / "a.ply", p)
    r = io.read_ply(tmp_path / "a.ply")
    assert (r[:, :3] == p).all() and (r[:, 3] == 1).all()
    # extra float properties are skipped by stride (read_points.hpp:98-105)
    with open(tmp_path / "b.ply", "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\nproperty float scalar_intensity\nend_header\n")
        f.write(np.arange(12, dtype="<f4").tobytes())
    assert (io.read_ply(tmp_path / "b.ply")[:, :3] == np.array([[0, 1, 2], [4, 5, 6], [8, 9, 10]], np.float32)).all()


def test_errors_are_reported_not_raised(tmp_path, capsys):
    assert io.read_ply(tmp_path / "missing.ply").shape == (0, 4) and io.read_points(tmp_path / "missing.bin").shape == (0, 4)
    with open(tmp_path / "c.ply", "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 1\nproperty double x\nend_header\n")
    assert io.read_ply(tmp_path / "c.ply").shape == (0, 4)
    err = capsys.readouterr().err
    assert "failed to open" in err and "only float properties are supported" in err


def test_reference_ply_matches_committed_points():
    """The reference's data/target.ply read by this reader == the xyz committed under tests/golden (when the reference tree is present)."""
    ply = "/root/reference/data/target.ply"
    if not os.path.exists(ply):
        import pytest

        pytest.skip("reference tree not present")
    pts = io.read_ply(ply)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "c1_points.npz"))
    assert pts.shape == (69088, 4) and (pts[:, :3] == gold["target"]).all()


def test_trajectory_format(tmp_path):
    T = np.eye(4)
    T[:3, 3] = [1.23456789, -2.0, 3.5]
    io.write_trajectory(tmp_path / "t.txt", [np.eye(4), T])
    lines = open(tmp_path / "t.txt").read().splitlines()
    assert lines[0] == "1.000000 0.000000 0.000000 0.000000 0.000000 1.000000 0.000000 0.000000 0.000000 0.000000 1.000000 0.000000"
    assert lines[1].split()[3] == "1.234568" and len(io.read_trajectory(tmp_path / "t.txt")) == 2
    names = [tmp_path / "000002.bin", tmp_path / "000000.bin", tmp_path / "x.txt"]
    for n in names:
        n.write_bytes(b"")
    assert [os.path.basename(x) for x in io.list_kitti_scans(tmp_path)] == ["000000.bin", "000002.bin"]
