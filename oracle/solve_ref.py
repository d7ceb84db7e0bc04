"""CPU oracle of the batched hyperbolic position solve (TEST INFRASTRUCTURE ONLY: imported by tests/,
never by the product).

Objective -- exactly the reference's (``tdoa_processor.py:249-273``), per window::

    f(p) = sum_k  w_k * ( |p - b2_k| - |p - b1_k| - d_k )^2 ,   w_k = 1 / (confidence_k + 0.1)

started from the centroid of the buoys' ECEF positions (``tdoa_processor.py:275-280``).  The reference
minimises it with scipy's BFGS one window at a time and reports ``sqrt(f_min / n)`` as accuracy
(``:300``); its own result is ill-determined along the poorly observed direction (all buoys sit
near one plane), which is why SURVEY.md section 8f ranks parity for this row as loose.  The batched
solver -- GPU product and this restatement alike -- is a fixed Levenberg-Marquardt rule instead
(documented in include/rmx.h: rmx_solve_batch), so that it is deterministic and comparable:

    lambda_0 = 1e-3;  each iteration: A = J^T J, g = J^T r, solve (A + lambda diag(A)) delta = -g
    (Cholesky, float64); accept when f decreases (lambda /= 3, floor 1e-12) else lambda *= 4;
    stop when an accepted |delta| < 1e-4 m, or lambda > 1e12, or after max_iter iterations.
"""
from __future__ import annotations

import numpy as np

SPEED_OF_LIGHT = 299792458.0  # tdoa_processor.py:141


def cost(p, b1, b2, d, w):
    r = np.linalg.norm(p - b2, axis=1) - np.linalg.norm(p - b1, axis=1) - d
    return float(np.sum(w * r * r))


def solve_one(buoy_xyz, pairs, d, w, max_iter=60):
    """buoy_xyz [B][3] float64, pairs [P][2] (buoy1, buoy2), d [P] metres (buoy2 - buoy1), w [P].
    Returns (p [3], f_min, iterations)."""
    b1, b2 = buoy_xyz[pairs[:, 0]], buoy_xyz[pairs[:, 1]]
    p = buoy_xyz.mean(axis=0)
    lam = 1e-3
    f = cost(p, b1, b2, d, w)
    it = 0
    while it < max_iter:
        it += 1
        v1, v2 = p - b1, p - b2
        n1, n2 = np.linalg.norm(v1, axis=1), np.linalg.norm(v2, axis=1)
        r = n2 - n1 - d
        J = v2 / n2[:, None] - v1 / n1[:, None]
        A = (J * w[:, None]).T @ J
        g = (J * w[:, None]).T @ r
        Ad = A + lam * np.diag(np.diag(A))
        try:
            delta = -np.linalg.solve(Ad, g)
        except np.linalg.LinAlgError:
            lam *= 4.0
            if lam > 1e12:
                break
            continue
        pn = p + delta
        fn = cost(pn, b1, b2, d, w)
        if fn < f:
            p, f = pn, fn
            lam = max(lam / 3.0, 1e-12)
            if float(np.linalg.norm(delta)) < 1e-4:
                break
        else:
            lam *= 4.0
            if lam > 1e12:
                break
    return p, f, it


def solve_batch(buoy_xyz, pairs, dist_diff, weight=None, max_iter=60):
    buoy_xyz = np.asarray(buoy_xyz, np.float64)
    pairs = np.asarray(pairs, np.int64).reshape(-1, 2)
    dist_diff = np.asarray(dist_diff, np.float64)
    W, P = dist_diff.shape
    weight = np.ones((W, P)) if weight is None else np.asarray(weight, np.float64)
    pos = np.zeros((W, 3))
    fmin = np.zeros(W)
    iters = np.zeros(W, np.int32)
    for i in range(W):
        pos[i], fmin[i], iters[i] = solve_one(buoy_xyz, pairs, dist_diff[i], weight[i], max_iter)
    return pos, fmin, iters


def lags_to_dist(lag_int, lag_frac, sample_rate_hz):
    """S7: distance difference in metres of a lag in samples (tdoa_processor.py:169-170)."""
    return (np.asarray(lag_int, np.float64) + np.asarray(lag_frac, np.float64)) / sample_rate_hz * SPEED_OF_LIGHT
