"""Frozen synthetic workloads for the BASELINE.json configs.  The reference ships no data at these
sizes, so these generators ARE the definition of C2-C5; they are deterministic in (n, seed).

scene(n, seed): LiDAR-like mixture of planar patches in a 100 m x 100 m x 15 m volume —
  40 % ground plane z = 0 over [-50,50]^2, 50 % on 30 axis-aligned vertical walls of random extent 2-20 m and height
  2-10 m, 10 % uniform clutter in [-50,50]^2 x [0,5]; Gaussian noise sigma = 0.01 m; float32.
The wall layout depends only on `layout_seed`, so two calls with different `seed` resample the SAME surfaces.
"""
import numpy as np


def _rot(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def gt_transform():
    """T_target_source of C2/C3/C4: 2 deg about (0.2, 0.3, 0.93) and t = (0.30, -0.20, 0.05)."""
    T = np.eye(4)
    T[:3, :3] = _rot([0.2, 0.3, 0.93], np.deg2rad(2.0))
    T[:3, 3] = [0.30, -0.20, 0.05]
    return T


def scene(n, seed, layout_seed=12345, noise=0.01):
    lay = np.random.default_rng(layout_seed)
    nwalls = 30
    centers = lay.uniform(-45, 45, size=(nwalls, 2))
    lengths = lay.uniform(2, 20, size=nwalls)
    heights = lay.uniform(2, 10, size=nwalls)
    along_x = lay.integers(0, 2, size=nwalls).astype(bool)
    rng = np.random.default_rng(seed)
    n_ground = int(0.4 * n)
    n_clutter = int(0.1 * n)
    n_wall = n - n_ground - n_clutter
    ground = np.stack([rng.uniform(-50, 50, n_ground), rng.uniform(-50, 50, n_ground), np.zeros(n_ground)], axis=1)
    area = lengths * heights
    which = rng.choice(nwalls, size=n_wall, p=area / area.sum())
    u = rng.uniform(-0.5, 0.5, n_wall) * lengths[which]
    v = rng.uniform(0, 1, n_wall) * heights[which]
    wx = np.where(along_x[which], centers[which, 0] + u, centers[which, 0])
    wy = np.where(along_x[which], centers[which, 1], centers[which, 1] + u)
    walls = np.stack([wx, wy, v], axis=1)
    clutter = np.stack([rng.uniform(-50, 50, n_clutter), rng.uniform(-50, 50, n_clutter), rng.uniform(0, 5, n_clutter)], axis=1)
    pts = np.concatenate([ground, walls, clutter], axis=0)
    pts += rng.normal(0, noise, size=pts.shape)
    pts = pts[rng.permutation(len(pts))]
    return np.ascontiguousarray(pts, dtype=np.float32)


def registration_pair(n, target_seed=1, source_seed=2):
    """(target, source, T_gt): independent resamples of one scene; the source is expressed in a frame displaced by T_gt^-1."""
    T = gt_transform()
    target = scene(n, target_seed)
    src_world = scene(n, source_seed).astype(np.float64)
    Ti = np.linalg.inv(T)
    source = (src_world @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
    return target, source, T


def kitti_like_scan(frame, n_rings=64, n_az=2030, layout_seed=12345, noise=0.02, seed=777):
    """C5: one ~130k-return scan of the scene from a sensor 1.8 m above ground moving 1 m/frame with 1 deg/frame yaw.
    Rays are cast against the ground plane and the 30 walls analytically; returns beyond 80 m or closer than 3 m are dropped.
    Returns (points in the sensor frame float32 (N,3), T_world_sensor)."""
    lay = np.random.default_rng(layout_seed)
    nwalls = 30
    centers = lay.uniform(-45, 45, size=(nwalls, 2))
    lengths = lay.uniform(2, 20, size=nwalls)
    heights = lay.uniform(2, 10, size=nwalls)
    along_x = lay.integers(0, 2, size=nwalls).astype(bool)
    yaw = np.deg2rad(1.0 * frame)
    # drive on a gentle arc starting near the scene centre
    pos = np.array([-30.0, -20.0, 1.8])
    for f in range(frame):
        a = np.deg2rad(1.0 * f)
        pos = pos + np.array([np.cos(a), np.sin(a), 0.0])
    Tws = np.eye(4)
    Tws[:3, :3] = _rot([0, 0, 1], yaw)
    Tws[:3, 3] = pos
    el = np.deg2rad(np.linspace(-24.8, 2.0, n_rings))
    az = np.linspace(-np.pi, np.pi, n_az, endpoint=False)
    EL, AZ = np.meshgrid(el, az, indexing="ij")
    d_s = np.stack([np.cos(EL) * np.cos(AZ), np.cos(EL) * np.sin(AZ), np.sin(EL)], axis=-1).reshape(-1, 3)
    d = d_s @ Tws[:3, :3].T
    o = pos
    t_best = np.full(len(d), np.inf)
    # ground z = 0
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = -o[2] / d[:, 2]
        hit = o + tg[:, None] * d
    ok = (tg > 0) & np.isfinite(tg)
    ok &= (np.abs(hit[:, 0]) <= 50) & (np.abs(hit[:, 1]) <= 50)
    t_best = np.where(ok, tg, t_best)
    for w in range(nwalls):
        ax = 1 if along_x[w] else 0  # wall plane is constant in this axis
        other = 1 - ax
        with np.errstate(divide="ignore", invalid="ignore"):
            tw = (centers[w, ax] - o[ax]) / d[:, ax]
            hitw = o + tw[:, None] * d
        okw = (tw > 0) & np.isfinite(tw) & (np.abs(hitw[:, other] - centers[w, other]) <= 0.5 * lengths[w]) & (hitw[:, 2] >= 0) & (hitw[:, 2] <= heights[w])
        t_best = np.where(okw & (tw < t_best), tw, t_best)
    keep = np.isfinite(t_best) & (t_best >= 3.0) & (t_best <= 80.0)
    rng = np.random.default_rng(seed + frame)
    r = t_best[keep] + rng.normal(0, noise, keep.sum())
    pts = d_s[keep] * r[:, None]
    return np.ascontiguousarray(pts, dtype=np.float32), Tws
