"""Test helper: the 96-double accumulator of the engine (csrc/linearize.hip: row layout) assembled on the CPU from the oracle's factor
state — [0, 29) the system, [32, 41) sum p_a g_j, [41, 59) sum p_a M'_c, [59, 95) sum p_a p_b M'_c with M' = R^T M R and g = R^T M r
(r = target - T source) over the accepted pairs; c = xx, xy, xz, yy, yz, zz, pairs ab = 00, 01, 02, 11, 12, 22."""
import numpy as np


def accumulator96(H, b, e, num_inliers, T, source_xyz, target_xyz, target_index, maha):
    acc = np.zeros(96)
    k = 0
    for i in range(6):
        for j in range(i, 6):
            acc[k] = H[i, j]
            k += 1
    acc[21:27] = b
    acc[27] = e
    acc[28] = num_inliers
    ok = np.asarray(target_index) >= 0
    p = np.asarray(source_xyz, dtype=np.float64)[ok]
    t = np.asarray(target_xyz, dtype=np.float64)[np.asarray(target_index)[ok]]
    M = np.asarray(maha, dtype=np.float64)[ok]
    R, tau = np.asarray(T)[:3, :3], np.asarray(T)[:3, 3]
    r = t - (p @ R.T + tau)
    Mp = np.einsum("ji,njk,kl->nil", R, M, R)          # R^T M R
    g = np.einsum("ji,njk,nk->ni", R, M, r)            # R^T M r
    sym = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
    m6 = np.stack([Mp[:, a, c] for a, c in sym], axis=1)
    for a in range(3):
        acc[32 + 3 * a:35 + 3 * a] = (p[:, a:a + 1] * g).sum(0)
        acc[41 + 6 * a:47 + 6 * a] = (p[:, a:a + 1] * m6).sum(0)
    for k, (a, c) in enumerate(sym):
        acc[59 + 6 * k:65 + 6 * k] = ((p[:, a] * p[:, c])[:, None] * m6).sum(0)
    return acc
