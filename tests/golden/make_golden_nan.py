#!/usr/bin/env python3
"""Fixture G10: the NaN-Mahalanobis case (reference gsplatcu/kernel.cu:243-246).  Since round 5 the build's default
follows the CUDA extension for a Gaussian whose conic HOLDS a NaN (EgsPolicy.nan_maha = 0, oracle mode "entry"); what
is left of the deviation is the inf * 0 pixel column of an infinite conic, and the opt-in "skip" policy.

    float maha_dist = max(0.0f, mahaSqDist(cinv, d));              // CUDA: max(0.f, NaN) == 0.f
    float alpha_prime = min(0.99f, alpha * exp(-0.5f * maha_dist)); //  -> min(0.99, alpha): the Gaussian BLENDS

A conic with a non-finite entry that survived inverseCov2D's NaN-determinant cull (cinv = (inf, 0, c): det = 0
exactly; or a caller-made NaN) gives NaN for every pixel with 0 * inf or NaN in the sum.  The CUDA extension then
paints such pixels with min(0.99, alpha) of the Gaussian's colour (and its backward pass produces NaN gradients from
cinv * d); this build SKIPS them in both draw kernels -- no NaN reaches the image or any gradient.

No CUDA device exists in this environment and neither of the reference's CPU scripts reaches the case (their conics
come from well-conditioned covariances), so BOTH expected images here are produced by the repo's oracle
(oracle/gs_oracle.py, NAN_MAHA = "skip" / "cuda"); the "cuda" one restates IEEE-754 / CUDA fmaxf semantics and is
what the test shows this build to differ from.  Inputs are 2D Gaussians handed straight to ``splat``.
    python tests/golden/make_golden_nan.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import gs_oracle as O   # noqa: E402

W, H = 32, 16
us = np.array([[7.3, 8.1], [20.0, 6.0], [12.5, 9.5], [25.2, 10.7]], np.float32)
cinv = np.array([[0.08, 0.01, 0.06], [np.inf, 0.0, 0.05], [np.nan, 0.0, 0.04], [0.05, -0.02, 0.09]], np.float32)
alphas = np.array([0.7, 0.6, 0.5, 0.9], np.float32)
colors = np.array([[0.9, 0.2, 0.1], [0.1, 0.8, 0.3], [0.2, 0.3, 0.9], [0.5, 0.5, 0.1]], np.float32)
depths = np.array([2.0, 1.0, 1.5, 3.0], np.float32)
areas = np.array([[12, 14], [5, 9], [9, 9], [13, 10]], np.int32)
out = dict(width=W, height=H, us=us, cinv2ds=cinv, alphas=alphas, colors=colors, depths=depths, areas=areas)
for mode in ("skip", "entry", "cuda"):
    O.NAN_MAHA = mode
    with np.errstate(all="ignore"):
        img, cont, tau, ranges, gsid = O.splat(H, W, us, cinv, alphas.astype(np.float64), depths.copy(), colors,
                                               areas.copy(), O.POLICY_G)
    out["image_" + mode] = img.astype(np.float32)
    out["contrib_" + mode] = cont
    out["tau_" + mode] = tau.astype(np.float32)
O.NAN_MAHA = "skip"
out["ranges"] = ranges; out["gsid"] = gsid
assert all(np.isfinite(out["image_" + m]).all() for m in ("skip", "entry", "cuda"))
d = np.abs(out["image_skip"] - out["image_cuda"]).max(0)
print("pixels that differ, skip vs cuda:", int((d > 1e-6).sum()), "of", W * H, "max", d.max())
d = np.abs(out["image_entry"] - out["image_cuda"]).max(0)
print("pixels that differ, entry vs cuda:", int((d > 1e-6).sum()), "columns", np.unique(np.nonzero(d > 1e-6)[1]))
O.NAN_MAHA = "cuda"
from tests.golden import _recipe   # noqa: E402
_recipe.begin("--check" in sys.argv[1:])
_recipe.save("g10_nan_conic.npz", "G10: oracle/gs_oracle.py splat on four 2D Gaussians, two of them with a non-finite "
             "conic: NAN_MAHA = 'cuda' (fmaxf(0, NaN) = 0, kernel.cu:243-246), 'entry' (the build's default: the same for the "
             "NaN conic, the inf conic's inf * 0 pixels skipped) and 'skip' (opt-in policy gsplatcu_nan_skip)", **out)
sys.exit(_recipe.finish())
