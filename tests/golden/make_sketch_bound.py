"""Golden vectors for the one test the reference holds of its second-order (sketched LM) path:
/root/reference/tests/test_sketching.py + tests/sketch_utils.py (CountSketch + damped lstsq on an
m x 8 problem, ||x_opt - x_sketch|| below two bounds).

Run in the build container only (the reference tree does not exist on the GPU box):
    python tests/golden/make_sketch_bound.py
It IMPORTS the reference's sketch_utils (numpy / scipy only), seeds numpy and stores in
tests/golden/sketch_bound.npz what its functions return with the parameters of
test_sketching.py:8-18 for m = 160*120 and m = 640*480:

  * gen_problem              A (fp32; every row for the small m, every 64th row + column sums for the
                             large one), b, x, and lstsq's x_opt / residual / sigma_min on the damped system
  * get_sketching_matrix     the bucket of every row (argmax of the reference's S), stack 1 / sketch 32
  * get_distortion           of the reference's own sketched system
  * the two bounds of run_test (formula of sketch_utils.py:84-98 evaluated here on the reference's
    outputs) and x_sketch; run_test itself is also run on the same seed: its asserts must hold.

Only arrays are stored; no reference source is copied.
"""
import io
import math
import os
import sys
from contextlib import redirect_stdout

import numpy as np
from scipy.linalg import lstsq

sys.path.insert(0, "/root/reference/tests")
import sketch_utils as R  # noqa: E402

P = dict(n=8, noise=1e-5, lambda_=10000.0, x_norm=0.015, max_sigma=5.0, min_sigma=1e-2)   # test_sketching.py:8-14
REPEAT, STACK, SKETCH = 1, 1, 32                                                          # :15-17
out = {"params": np.array([P["n"], P["noise"], P["lambda_"], P["x_norm"], P["max_sigma"], P["min_sigma"]])}

for tag, m, seed in (("small", 160 * 120, 11), ("large", 640 * 480, 12)):
    np.random.seed(seed)
    A, A_damp, b, b_damp, x = R.gen_problem(m, P["n"], lambda_=P["lambda_"], noise=P["noise"], x_norm=P["x_norm"],
                                            max_sigma=P["max_sigma"], min_sigma=P["min_sigma"])
    x_opt = lstsq(A_damp, b_damp)[0]
    res = np.linalg.norm(A_damp @ x_opt - b_damp, 2)
    S = R.get_sketching_matrix(m, P["n"], REPEAT, STACK, SKETCH, mode="count")
    d = REPEAT * STACK * SKETCH
    A_t = np.vstack([S @ A, math.sqrt(P["lambda_"]) * np.eye(P["n"])])
    b_t = np.concatenate([S @ b, np.zeros(P["n"])])
    x_sketch = lstsq(A_t, b_t)[0]
    res_sketch = np.linalg.norm(A_damp @ x_sketch - b_damp, 2)
    dist = R.get_distortion(A_damp, A_t)
    dist_hat = math.sqrt(P["n"] / d)
    smin = np.linalg.svd(A_damp, compute_uv=False)[-1]
    smin_hat = np.linalg.svd(A_t, compute_uv=False)[-1]
    gamma, gamma_hat = (1 + dist) / (1 - dist), (1 + dist_hat) / (1 - dist_hat)
    ub = res * math.sqrt(gamma ** 2 - 1) / smin
    ub_hat = res_sketch * gamma * math.sqrt(gamma_hat ** 2 - 1) / smin_hat
    assert np.linalg.norm(x_opt - x_sketch) < min(ub, ub_hat)
    out[f"{tag}_seed"] = np.array(seed)
    out[f"{tag}_m"] = np.array(m)
    if tag == "small":
        out["small_A"] = A.astype(np.float32)
        out["small_b"] = b.astype(np.float32)
    else:
        out["large_A_rows64"] = A[::64].astype(np.float32)
        out["large_b_rows64"] = b[::64].astype(np.float32)
    out[f"{tag}_A_colsum"] = A.sum(0)
    out[f"{tag}_AtA"] = A.T @ A
    out[f"{tag}_Atb"] = A.T @ b
    out[f"{tag}_x"] = x
    out[f"{tag}_x_opt"] = x_opt
    out[f"{tag}_res"] = np.array(res)
    out[f"{tag}_sigma_min"] = np.array(smin)
    out[f"{tag}_ref_bucket"] = S.argmax(0).astype(np.int16)
    out[f"{tag}_ref_x_sketch"] = x_sketch
    out[f"{tag}_ref_distortion"] = np.array(dist)
    out[f"{tag}_ref_bounds"] = np.array([ub, ub_hat])
    out[f"{tag}_ref_x_diff"] = np.array(np.linalg.norm(x_opt - x_sketch))
    if tag == "small":
        # The first bound is a heuristic (its distortion looks at the two extreme singular values only) and
        # the reference's own sketch misses it now and then: 30 more of the reference's sketches of the SAME
        # problem, so that a test can hold another sketch to the reference's own statistics.
        diffs, ubs, ubhs = [], [], []
        for s2 in range(30):
            np.random.seed(1000 + s2)
            S2 = R.get_sketching_matrix(m, P["n"], REPEAT, STACK, SKETCH, mode="count")
            A2 = np.vstack([S2 @ A, math.sqrt(P["lambda_"]) * np.eye(P["n"])])
            b2 = np.concatenate([S2 @ b, np.zeros(P["n"])])
            x2 = lstsq(A2, b2)[0]
            d2 = R.get_distortion(A_damp, A2)
            g2 = (1 + d2) / (1 - d2)
            diffs.append(np.linalg.norm(x_opt - x2))
            ubs.append(res * math.sqrt(g2 ** 2 - 1) / smin)
            ubhs.append(np.linalg.norm(A_damp @ x2 - b_damp, 2) * g2 * math.sqrt(gamma_hat ** 2 - 1) /
                        np.linalg.svd(A2, compute_uv=False)[-1])
        out["small_ref30_x_diff"] = np.array(diffs)
        out["small_ref30_bound"] = np.array(ubs)
        out["small_ref30_bound_hat"] = np.array(ubhs)
        print("reference sketch, 30 draws on the small problem: first bound met by",
              int((np.array(diffs) < np.array(ubs)).sum()), "second by", int((np.array(diffs) < np.array(ubhs)).sum()),
              f"median x_diff {np.median(diffs):.3e}")
    # the reference's own run_test on the same seed: it draws the same problem and sketch and asserts
    np.random.seed(seed)
    with redirect_stdout(io.StringIO()) as log:
        R.run_test(m, P["n"], P["noise"], P["x_norm"], P["lambda_"], P["max_sigma"], P["min_sigma"], REPEAT, STACK,
                   SKETCH, "append_damp", "count")
    print(tag, "reference run_test passed;", " | ".join(log.getvalue().strip().splitlines()[6:8]),
          f"| x_diff {np.linalg.norm(x_opt - x_sketch):.3e} bounds {ub:.3e} {ub_hat:.3e}")

dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sketch_bound.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, os.path.getsize(dst), "bytes")
