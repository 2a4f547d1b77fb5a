"""Row b (drop-in boundary), host side, no GPU: the import statements of the reference's OWN callers must resolve
with this repository on ``PYTHONPATH``.

``forward_gpu.py:2-6`` does ``import gsplatcu as gsc`` + ``from gsplat.gau_io import *`` + ``from gsplat.gausplat
import *``; ``backward_gpu.py:3-7`` adds ``from gsplat.sh_coef import *``.  The reference's ``gsplat`` is a namespace
package (a directory without ``__init__.py``) next to the caller.  Round 3 shipped a regular ``gsplat/`` package at the
repository root, which SHADOWED it: ``gsplat.gausplat`` / ``gsplat.sh_coef`` raised ``ModuleNotFoundError``.  Now

  * the repository root provides ``gsplatcu`` only: the caller's whole ``gsplat`` stays its own;
  * ``<repo>/compat`` (opt-in) provides the MI355X ``gsplat.gsmodel`` / ``gau_io`` / ... and extends
    its package path, so the caller's ``gsplat.gausplat`` / ``gsplat.sh_coef`` are still found.

The test builds a stand-in caller tree in a temp dir (own stub modules, a three-line caller: no reference file is
copied) and imports it in a subprocess; when ``/root/reference`` exists (build container only) the reference's real
``gsplat.sh_coef`` / ``gsplat.gausplat`` are imported the same way."""
import os
import subprocess
import sys
import textwrap

import pytest

from tests.conftest import REPO

REF = "/root/reference"


def _run(code, cwd, pythonpath):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join(pythonpath), PYTHONDONTWRITEBYTECODE="1", MPLBACKEND="Agg")
    return subprocess.run([sys.executable, "-c", textwrap.dedent(code)], cwd=cwd, env=env, capture_output=True,
                          text=True, timeout=300)


def _caller_tree(tmp_path):
    pkg = tmp_path / "caller" / "gsplat"                      # a namespace package, like the reference's
    pkg.mkdir(parents=True)
    (pkg / "sh_coef.py").write_text("MARK_SH = 'caller sh_coef'\n")
    (pkg / "gausplat.py").write_text("MARK_GAUSPLAT = 'caller gausplat'\n")
    (pkg / "gau_io.py").write_text("MARK_IO = 'caller gau_io'\n")
    (tmp_path / "caller" / "caller.py").write_text(textwrap.dedent("""
        import gsplatcu as gsc
        from gsplat.gau_io import *
        from gsplat.gausplat import *
        from gsplat.sh_coef import *
        import gsplat, sys
        print("GSC", gsc.__file__)
        print("OPS", all(callable(getattr(gsc, n)) for n in
                         ("project", "computeCov3D", "computeCov2D", "sh2Color", "inverseCov2D", "splat", "splatB")))
        print("SH", MARK_SH); print("GAUSPLAT", MARK_GAUSPLAT)
        print("IO", sys.modules["gsplat.gau_io"].__file__)
    """))
    return str(tmp_path / "caller")


def _fields(out):
    return dict(ln.split(" ", 1) for ln in out.strip().splitlines() if " " in ln)


def test_plain_dropin_keeps_the_callers_gsplat_package(tmp_path):
    cwd = _caller_tree(tmp_path)
    r = _run("import runpy; runpy.run_path('caller.py', run_name='__main__')", cwd, [cwd, REPO])
    assert r.returncode == 0, r.stderr[-2000:]
    f = _fields(r.stdout)
    assert f["GSC"] == os.path.join(REPO, "gsplatcu", "__init__.py") and f["OPS"] == "True"
    assert f["SH"] == "caller sh_coef" and f["GAUSPLAT"] == "caller gausplat"
    assert f["IO"].startswith(cwd)                      # the caller's own gau_io: nothing of gsplat.* is replaced


def test_opt_in_compat_replaces_some_names_and_hides_none(tmp_path):
    cwd = _caller_tree(tmp_path)
    r = _run("import runpy; runpy.run_path('caller.py', run_name='__main__')", cwd,
             [os.path.join(REPO, "compat"), cwd, REPO])
    assert r.returncode == 0, r.stderr[-2000:]
    f = _fields(r.stdout)
    assert f["GSC"] == os.path.join(REPO, "gsplatcu", "__init__.py")
    assert f["SH"] == "caller sh_coef" and f["GAUSPLAT"] == "caller gausplat"      # not provided here: the caller's
    assert f["IO"] == os.path.join(REPO, "compat", "gsplat", "gau_io.py")           # provided here: replaced


def test_repository_root_has_no_gsplat_package():
    assert not os.path.exists(os.path.join(REPO, "gsplat"))


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout only exists in the build container")
@pytest.mark.parametrize("compat", [False, True])
def test_reference_modules_resolve_next_to_the_dropin(compat):
    """The real thing: the reference's gsplat.sh_coef / gsplat.gausplat (forward_gpu.py:6, backward_gpu.py:5) import
    from its checkout with this repository (and optionally compat/) on PYTHONPATH; gsplatcu is this repository's."""
    path = ([os.path.join(REPO, "compat")] if compat else []) + [REF, REPO]
    r = _run("""
        import gsplatcu as gsc
        from gsplat.sh_coef import *
        from gsplat.gausplat import *
        import sys
        print("GSC", gsc.__file__)
        print("SH", sys.modules["gsplat.sh_coef"].__file__)
        print("GAUSPLAT", sys.modules["gsplat.gausplat"].__file__)
        print("C0", SH_C0_0)
    """, "/tmp", path)
    assert r.returncode == 0, r.stderr[-2000:]
    f = _fields(r.stdout)
    assert f["GSC"].startswith(REPO) and f["SH"].startswith(REF) and f["GAUSPLAT"].startswith(REF)
    assert abs(float(f["C0"]) - 0.28209479177387814) < 1e-12
