"""On-disk formats either side of the registration path.

* read_points / write_points : the KITTI `.bin` layout, a flat little-endian float32 x,y,z,intensity array
  (include/small_gicp/benchmark/read_points.hpp:15-46 of the reference; the 4th component is replaced by 1).
* read_ply                   : "simple PLY" = binary_little_endian, one `element vertex N`, float properties only, first three
  x, y, z (read_points.hpp:52-109; the reference rejects any other property type and warns when the first three are not x/y/z).
* list_kitti_scans           : sorted `*.bin` files of a directory (benchmark.hpp:96-115).
* write_trajectory           : one line per pose, the 3x4 top rows as 12 numbers "%.6f" separated by blanks
  (src/benchmark/odometry_benchmark.cpp:82-94).
"""
import os
import sys

import numpy as np


def read_points(filename):
    """(N,4) float32, w = 1; empty array (and a message on stderr) if the file cannot be opened — the reference's behaviour."""
    try:
        raw = np.fromfile(filename, dtype="<f4")
    except OSError:
        print("error: failed to open %s" % filename, file=sys.stderr)
        return np.zeros((0, 4), np.float32)
    pts = raw[: (len(raw) // 4) * 4].reshape(-1, 4).astype(np.float32, copy=True)
    pts[:, 3] = 1.0
    return pts


def write_points(filename, points):
    p = np.asarray(points, dtype=np.float32)
    if p.shape[1] == 3:
        p = np.concatenate([p, np.ones((len(p), 1), np.float32)], axis=1)
    p.astype("<f4").tofile(filename)


def read_ply(filename):
    """(N,4) float32 with w = 1, or an empty array after an error message (never raises: read_points.hpp:52-109)."""
    try:
        f = open(filename, "rb")
    except OSError:
        print("error: failed to open %s" % filename, file=sys.stderr)
        return np.zeros((0, 4), np.float32)
    with f:
        props, n = [], 0
        while True:
            line = f.readline()
            if not line:
                break
            line = line.decode("ascii", "replace").rstrip("\r\n")
            if not line or line == "end_header":
                break
            tok = line.split()
            if tok[0] == "element":
                if len(tok) < 3 or tok[1] != "vertex":
                    print("error: invalid ply format (line=%s)" % line, file=sys.stderr)
                    return np.zeros((0, 4), np.float32)
                n = int(tok[2])
            elif tok[0] == "property":
                if len(tok) < 3 or tok[1] != "float":
                    print("error: only float properties are supported!! (line=%s)" % line, file=sys.stderr)
                    return np.zeros((0, 4), np.float32)
                props.append(tok[2])
        if len(props) < 3 or [p.lower() for p in props[:3]] != ["x", "y", "z"]:
            print("warning: invalid properties!!", file=sys.stderr)
            for p in props:
                print(" - %s" % p, file=sys.stderr)
            if len(props) < 3:
                return np.zeros((0, 4), np.float32)
        data = np.frombuffer(f.read(4 * len(props) * n), dtype="<f4")
    n = min(n, len(data) // len(props))
    data = data[: n * len(props)].reshape(n, len(props))
    pts = np.ones((n, 4), np.float32)
    pts[:, :3] = data[:, :3]
    return pts


def write_ply(filename, points):
    """Writer for the same simple layout (x, y, z float32) — used by the tests and to export clouds."""
    p = np.asarray(points, dtype=np.float32)[:, :3]
    with open(filename, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nend_header\n" % len(p)).encode("ascii"))
        f.write(np.ascontiguousarray(p, dtype="<f4").tobytes())


def list_kitti_scans(dataset_path, max_num_data=1000000):
    names = sorted(os.path.join(dataset_path, f) for f in os.listdir(dataset_path) if f.endswith(".bin"))
    return names[:max_num_data]


def write_trajectory(filename, poses):
    with open(filename, "w") as f:
        for T in poses:
            T = np.asarray(T, dtype=np.float64)
            f.write(" ".join("%.6f" % T[i, j] for i in range(3) for j in range(4)) + "\n")


def read_trajectory(filename):
    out = []
    for line in open(filename):
        v = [float(x) for x in line.split()]
        if len(v) == 12:
            T = np.eye(4)
            T[:3, :4] = np.asarray(v).reshape(3, 4)
            out.append(T)
    return out
