"""Shared by the fixture recipes (make_golden*.py): where a fixture is written, the ``--check`` mode and the guard that
what was imported IS the reference.

``--check`` regenerates every fixture of a recipe into a temporary directory and compares it, array by array, with the
committed file: names, shapes, dtypes, values (integers / strings exact; floats to 1e-12 relative -- BLAS summation
order is the only freedom between two runs of the same NumPy code).  ``tests/test_golden_recipes.py`` runs it for
every recipe when ``/root/reference`` exists, so the parity pin stays reproducible from HEAD."""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
_check_dir = None
_written = []


def begin(check):
    global _check_dir
    if check:
        _check_dir = tempfile.mkdtemp(prefix="egs_golden_check_")


def assert_reference(*mods):
    """Every module the fixtures are generated FROM must come from the reference checkout -- not from this repository's
    same-named modules (compat/gsplat/*), which would make a 'golden' out of the code under test."""
    for m in mods:
        f = os.path.realpath(getattr(m, "__file__", "") or "")
        if not f.startswith(os.path.realpath(REF) + os.sep):
            raise RuntimeError("%s was imported from %r, not from %s" % (m.__name__, f, REF))


def save(name, doc, **arrays):
    path = os.path.join(_check_dir or HERE, name)
    np.savez_compressed(path, __doc__=np.array(doc), **arrays)
    _written.append(name)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


def finish():
    """In --check mode: compare what was written with the committed fixtures; exit status 1 on any difference."""
    if _check_dir is None:
        return 0
    bad = 0
    for name in _written:
        with np.load(os.path.join(_check_dir, name), allow_pickle=False) as new, \
                np.load(os.path.join(HERE, name), allow_pickle=False) as old:
            if sorted(new.files) != sorted(old.files):
                print("CHECK %s: array names differ: %s" % (name, sorted(set(new.files) ^ set(old.files)))); bad += 1
                continue
            worst = 0.0
            for k in new.files:
                a, b = new[k], old[k]
                if a.shape != b.shape or a.dtype != b.dtype:
                    print("CHECK %s[%s]: %s %s != committed %s %s" % (name, k, a.dtype, a.shape, b.dtype, b.shape))
                    bad += 1
                elif a.dtype.kind in "fc":
                    scale = max(1e-300, float(np.abs(b[np.isfinite(b)]).max())) if np.isfinite(b).any() else 1.0
                    same_nan = np.array_equal(np.isnan(a), np.isnan(b))
                    d = float(np.nanmax(np.abs(np.where(np.isfinite(a) & np.isfinite(b), a - b, 0.0)))) if a.size else 0.0
                    worst = max(worst, d / scale)
                    if not same_nan or d > 1e-12 * scale or not np.array_equal(np.isinf(a), np.isinf(b)):
                        print("CHECK %s[%s]: max|delta| = %.3g (scale %.3g)" % (name, k, d, scale)); bad += 1
                elif not np.array_equal(a, b):
                    print("CHECK %s[%s]: differs" % (name, k)); bad += 1
            print("CHECK %s: %d arrays, worst relative difference %.2g" % (name, len(new.files), worst))
    print("CHECK", "FAILED" if bad else "ok: regenerated fixtures == committed fixtures")
    import shutil
    shutil.rmtree(_check_dir, ignore_errors=True)
    return 1 if bad else 0
