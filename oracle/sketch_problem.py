"""CHECKER ONLY (test infrastructure): numpy restatement of the reference's own test of the
second-order (sketched Levenberg-Marquardt) solve, /root/reference/tests/sketch_utils.py and
tests/test_sketching.py.  Only tests/ may import this; the product never does.

Pinned: tests/golden/sketch_bound.npz holds what the REFERENCE's functions return on the same numpy
seeds (tests/golden/make_sketch_bound.py imports them in the build container);
tests/test_cpu_golden.py::test_sketch_problem_restatement_matches_the_reference checks this file
against those arrays.

  gen_problem      sketch_utils.py:5-21   (the damped least-squares problem of test_sketching.py:8-18)
  count_sketch     sketch_utils.py:23-34  (mode "count": unsigned, buckets drawn with replacement)
  distortion       sketch_utils.py:36-44
  bounds           sketch_utils.py:58-124 (what run_test asserts: ||x_opt - x_sketch|| below both)
"""
from __future__ import annotations

import math

import numpy as np

# tests/test_sketching.py:8-18
REFERENCE_TEST = dict(n=8, noise=1e-5, lambda_=10000.0, x_norm=0.015, max_sigma=5.0, min_sigma=1e-2)


def gen_problem(m, n=8, max_sigma=5.0, min_sigma=1e-2, lambda_=10000.0, noise=1e-5, x_norm=0.015, seed=0):
    """Same draws in the same order as gen_A + gen_problem (sketch_utils.py:5-21) after
    np.random.seed(seed): A = U diag(S) Vt with prescribed singular values, b = A x + noise."""
    rs = np.random.RandomState(seed)          # np.random.seed(seed) + the global functions draw the same stream
    A = rs.randn(m, n)
    U, _, Vt = np.linalg.svd(A, full_matrices=False)
    S = rs.uniform(min_sigma, max_sigma / 1.5, n)
    S[0], S[-1] = max_sigma, min_sigma
    A = U @ np.diag(S) @ Vt
    x = rs.randn(n)
    x = x_norm * x / np.linalg.norm(x)
    b = A @ x + noise * rs.randn(m)
    return A, b, x


def damped(A, b, lambda_):
    n = A.shape[1]
    return np.vstack([A, math.sqrt(lambda_) * np.eye(n)]), np.concatenate([b, np.zeros(n)])


def count_sketch(indices, rows, weights=None):
    """S [rows, m] with S[indices[p], p] = weights[p] (1 in the reference's test, +-1 in its tracker:
    rand_weights, slam_frontend.py:318); indices < 0 leave the column empty."""
    m = indices.shape[0]
    S = np.zeros((rows, m))
    keep = indices >= 0
    S[indices[keep], np.nonzero(keep)[0]] = 1.0 if weights is None else weights[keep]
    return S


def distortion(A, A_tilde):
    s = np.linalg.svd(A, compute_uv=False)
    st = np.linalg.svd(A_tilde, compute_uv=False)
    return max(abs(s[0] - st[0]) / s[0], abs(s[-1] - st[-1]) / s[-1])


def bounds(A, b, lambda_, SA, Sb, x_sketch, d):
    """The two right-hand sides run_test asserts against (sketch_utils.py:62-98, solve_mode
    "append_damp") for a sketched system (SA, Sb) of d rows and a candidate solution x_sketch.
    Returns (x_opt, upperbound, upperbound_hat, stats)."""
    from scipy.linalg import lstsq
    n = A.shape[1]
    A_damp, b_damp = damped(A, b, lambda_)
    x_opt = lstsq(A_damp, b_damp)[0]
    res = np.linalg.norm(A_damp @ x_opt - b_damp, 2)
    A_tilde, _ = damped(SA, Sb, lambda_)
    res_sketch = np.linalg.norm(A_damp @ x_sketch - b_damp, 2)
    dist = distortion(A_damp, A_tilde)
    dist_hat = math.sqrt(n / d)
    sigma_min = np.linalg.svd(A_damp, compute_uv=False)[-1]
    sigma_min_hat = np.linalg.svd(A_tilde, compute_uv=False)[-1]
    gamma = (1 + dist) / (1 - dist)
    gamma_hat = (1 + dist_hat) / (1 - dist_hat)
    ub = res * math.sqrt(gamma ** 2 - 1) / sigma_min
    ub_hat = res_sketch * gamma * math.sqrt(gamma_hat ** 2 - 1) / sigma_min_hat
    return x_opt, ub, ub_hat, dict(res=res, res_sketch=res_sketch, distortion=dist, sigma_min=sigma_min,
                                   sigma_min_hat=sigma_min_hat)
