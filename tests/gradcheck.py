"""Gradient acceptance rule of the parity tests -- RELATIVE to the gradient's own magnitude.

The reference checks gradients with ``|a-b| < 1e-4`` (backward_cpu.py:61-65) on values of O(1e-3 .. 1e-1)
(backward_gpu.py:126-152: L1 against zeros on a 32x16 image).  The tests here feed upstream gradients scaled by
1/(H*W), so every per-Gaussian gradient is far below 1 and an absolute 1e-4 (or ``1e-4*max(1,|ref|)``) accepts
anything, all-zeros included.  The rule below cannot be dodged by the scale of ``dl``:

  (1) max |got - ref|            <= tol_max * max|ref|                                  (2e-4)
  (2) over the entries with |ref| >= big * max|ref| (big = 1e-2; "the large entries"):
        median relative error    <= med_rel                                            (1e-4)
        relative error           <= max_rel   for all but ``outliers`` of them         (5e-3)
      the entries beyond max_rel are COUNTED (returned and asserted <= outliers, default 0; a float < 1 = that
      fraction of the large entries, for million-row tensors): they are where a
      pixel sits on the other side of an alpha' >= 0.002 / tau < 1e-4 threshold in fp32 (kernel.cu:246,256) --
      "threshold-flip Gaussians", reported separately as the image checks do with flipped pixels;
  (3) max|ref| > 0 (a comparison against an all-zero reference is vacuous and refused).

``report`` returns the numbers, ``assert_grad_close`` asserts them.  Negative controls
(tests/test_gradcheck_rule.py): a gradient scaled by 1.01, all-zeros, and a single wrong large entry all FAIL.
When the environment variable EGS_GRAD_STATS names a file, every call appends its numbers to it as a JSON line
(the tolerances above were set from such a dump taken on the GPU box, profiles/r4_grad_errors.jsonl)."""
import json
import os

import numpy as np

TOL_MAX, BIG, MED_REL, MAX_REL = 2e-4, 1e-2, 1e-4, 5e-3


def report(got, ref, big=BIG, max_rel=MAX_REL):
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    if got.shape != ref.shape:
        raise AssertionError("shape %s != reference shape %s" % (got.shape, ref.shape))
    if ref.size == 0:
        return dict(n=0, ref_max=0.0, abs_over_max=0.0, n_big=0, med_rel=0.0, worst_rel=0.0, n_out=0, finite=True)
    finite = bool(np.isfinite(got).all())
    rmax = float(np.abs(ref).max())
    err = np.abs(got - ref)
    sel = np.abs(ref) >= big * rmax if rmax > 0 else np.zeros(ref.shape, bool)
    rel = err[sel] / np.abs(ref[sel]) if sel.any() else np.zeros(0)
    return dict(n=int(ref.size), ref_max=rmax, abs_over_max=float(err.max() / rmax) if rmax > 0 else float("inf"),
                n_big=int(sel.sum()), med_rel=float(np.median(rel)) if rel.size else 0.0,
                worst_rel=float(rel.max()) if rel.size else 0.0, n_out=int((rel > max_rel).sum()), finite=finite)


def grad_close(got, ref, tol_max=TOL_MAX, big=BIG, med_rel=MED_REL, max_rel=MAX_REL, outliers=0):
    """-> (ok, report dict)."""
    r = report(got, ref, big, max_rel)
    # ``outliers``: an absolute count, or (a float below 1) a FRACTION of the large entries -- for million-row tensors,
    # where a handful of entries in a million sit on an unflagged threshold; at least 2 are then allowed
    allowed = max(2, int(outliers * r["n_big"])) if isinstance(outliers, float) and outliers < 1 else outliers
    r["outliers_allowed"] = int(allowed)
    ok = (r["finite"] and r["ref_max"] > 0 and r["abs_over_max"] <= tol_max and r["med_rel"] <= med_rel
          and r["n_out"] <= allowed)
    return ok, r


def assert_grad_close(got, ref, name="", **kw):
    ok, r = grad_close(got, ref, **kw)
    path = os.environ.get("EGS_GRAD_STATS")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps(dict(name=str(name), ok=bool(ok), **r)) + "\n")
        if os.environ.get("EGS_GRAD_STATS_ONLY"):
            return r
    assert ok, (name, r, kw)
    return r


def assert_grad_close_flips(got, ref, near, name="", near_frac=0.02, near_tol=2e-2, **kw):
    """The rule above on the Gaussians (rows) that are NOT ``near`` a skip threshold; the ``near`` rows -- where a
    float32 evaluation may blend or skip a pixel the float64 oracle treats the other way (``near_out`` of
    ``O.draw_backward``) -- are COUNTED (at most ``near_frac`` of the rows with a gradient) and bounded loosely:
    a flipped pixel moves a gradient by at most that pixel's own contribution, alpha' ~ 0.002 of a weight."""
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    near = np.asarray(near, bool)
    r = assert_grad_close(got[~near], ref[~near], name, **kw)
    has = np.abs(ref.reshape(ref.shape[0], -1)).max(1) > 0
    n_near = int((near & has).sum())
    err = float(np.abs(got[near] - ref[near]).max() / np.abs(ref).max()) if n_near else 0.0
    r["n_near"], r["n_rows"], r["near_err"] = n_near, int(has.sum()), err
    path = os.environ.get("EGS_GRAD_STATS")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps(dict(name=str(name) + ":near", n_near=n_near, n_rows=int(has.sum()), near_err=err)) + "\n")
        if os.environ.get("EGS_GRAD_STATS_ONLY"):
            return r
    assert n_near <= max(3, near_frac * has.sum()), (name, "threshold-flip Gaussians", n_near, int(has.sum()))
    assert err <= near_tol, (name, "threshold-flip Gaussians off by", err)
    return r
