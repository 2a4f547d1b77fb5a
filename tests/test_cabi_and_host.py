"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports
every symbol include/egs_hip.h declares, the ctypes mirror matches the C structs,
and the host-side validation of the Python op surface.  No GPU compute calls."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from tests.conftest import REPO

HEADER = os.path.join(REPO, "include", "egs_hip.h")


@pytest.fixture(scope="module")
def lib():
    from easygaussiansplatting_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(egs_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_seven_ops_and_cites_the_reference():
    names = declared_functions()
    for op in ("egs_project", "egs_cov3d", "egs_cov2d", "egs_sh2color", "egs_inv_cov2d", "egs_splat_bin",
               "egs_splat_draw", "egs_splat_bwd"):
        assert op in names
    txt = open(HEADER).read()
    for cite in ("ext.cpp:54-61", "ext.cpp:39-42", "ext.cpp:44-52", "ext.cpp:63-66", "ext.cpp:34-36",
                 "ext.cpp:20-32", "gausplat.cu:24-112"):
        assert cite in txt, cite


def test_library_exports_every_declared_symbol(lib):
    from easygaussiansplatting_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    for name in declared_functions():
        assert name in exported, name
        assert name in _lib.SIGNATURES, "ctypes signature missing for " + name
    assert lib.egs_abi_version() == _lib.ABI_VERSION


def test_no_torch_or_python_dependency_in_the_shared_library():
    from easygaussiansplatting_amd import _lib
    out = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
    needed = re.findall(r"NEEDED.*\[(.*?)\]", out)
    assert any("amdhip64" in n for n in needed)
    assert not any(("torch" in n) or ("python" in n) or ("c10" in n) for n in needed), needed


def test_policy_struct_layout_and_presets(lib):
    from easygaussiansplatting_amd._lib import EgsPolicy
    assert C.sizeof(EgsPolicy) == 52
    g = EgsPolicy(); lib.egs_policy_gsplatcu(C.byref(g))
    assert (g.near_cull, g.fov_mode, g.nan_cull, g.radius_mode, g.footprint, g.maha_floor, g.alpha_clamp,
            g.depth_key, g.nan_maha) == (1, 0, 1, 0, 0, 1, 1, 0, 0)
    assert abs(g.alpha_skip - 0.002) < 1e-9 and abs(g.tau_stop - 1e-4) < 1e-10 and g.det_eps == 0
    a = EgsPolicy(); lib.egs_policy_forward_cpu(C.byref(a))
    assert (a.near_cull, a.fov_mode, a.nan_cull, a.radius_mode, a.footprint, a.far_cull, a.maha_floor,
            a.alpha_clamp, a.depth_key, a.nan_maha) == (0, 1, 0, 1, 1, 1, 0, 1, 1, 1)
    assert abs(a.det_eps - 1e-6) < 1e-12 and a.alpha_skip == 0 and a.tau_stop == 0
    # the oracle's policies are the same two rows of SURVEY §8a-R0
    from oracle import gs_oracle as O
    for p, o in ((g, O.POLICY_G), (a, O.POLICY_A)):
        assert (bool(p.near_cull), p.fov_mode, bool(p.nan_cull), p.radius_mode, p.footprint, bool(p.maha_floor),
                bool(p.alpha_clamp)) == (o.near_cull, o.fov_mode, o.nan_cull, o.radius_mode, o.footprint,
                                         o.maha_floor, o.alpha_clamp)
        assert abs(p.alpha_skip - o.alpha_skip) < 1e-9 and abs(p.tau_stop - o.tau_stop) < 1e-9
        assert p.depth_key == o.depth_key


def test_argument_errors_are_reported_not_crashed(lib):
    from easygaussiansplatting_amd._lib import EgsPolicy
    p = EgsPolicy(); lib.egs_policy_gsplatcu(C.byref(p))
    rc = lib.egs_sh2color(4, 5, None, None, None, None, None, None, None)      # bad sh_dim
    assert rc == 10001 and b"bad argument" in lib.egs_last_error_string()
    rc = lib.egs_project(-1, None, None, None, 1., 1., 0., 0., C.byref(p), None, None, None, None, None)
    assert rc == 10001
    assert lib.egs_project(0, None, None, None, 1., 1., 0., 0., C.byref(p), None, None, None, None, None) == 0
    # workspace sizes are monotone and cover the documented layout
    assert lib.egs_splat_bin_ws_bytes(1_000_000) >= 1_000_000 * (2 * 8 + 5 * 4)   # 2 packed-rect + 5 u32 arrays
    assert lib.egs_splat_draw_ws_bytes(1_000_000, 4_100_000, 1920, 1080) >= 3 * 4 * 4_100_000 + 48 * 1_000_000
    assert lib.egs_splat_bin_ws_bytes(10) <= lib.egs_splat_bin_ws_bytes(1000)


def test_op_surface_validates_before_touching_the_gpu():
    torch = pytest.importorskip("torch")
    from easygaussiansplatting_amd import gsplatcu as gsc
    with pytest.raises(ValueError):      # CPU tensor
        gsc.project(torch.zeros(4, 3), torch.eye(3), torch.zeros(3), 1., 1., 0., 0., False)
    with pytest.raises(TypeError):
        gsc.project(np.zeros((4, 3), np.float32), torch.eye(3), torch.zeros(3), 1., 1., 0., 0., False)
    with pytest.raises(ValueError):
        gsc.splat(0, 16, torch.zeros(1, 2), torch.zeros(1, 3), torch.zeros(1), torch.zeros(1), torch.zeros(1, 3),
                  torch.zeros(1, 2, dtype=torch.int32))
    assert gsc.get_policy() in ("gsplatcu", "forward_cpu")
    import gsplatcu as top                 # the literal drop-in module name
    for f in ("project", "computeCov3D", "computeCov2D", "sh2Color", "inverseCov2D", "splat", "splatB"):
        assert callable(getattr(top, f))


def test_product_path_never_imports_the_oracle():
    """The package, the two drop-in shims and the example scripts; only tests/, bench.py's cpu_baseline leg and
    __graft_entry__.smoke() may touch oracle/."""
    for pkg in ("easygaussiansplatting_amd", "gsplatcu", "compat", "examples"):
        for root, _, files in os.walk(os.path.join(REPO, pkg)):
            for f in files:
                if f.endswith(".py"):
                    src = open(os.path.join(root, f)).read()
                    code = "\n".join(ln for ln in src.splitlines() if re.match(r"\s*(from|import)\s", ln))
                    assert "oracle" not in code, os.path.join(root, f)
    # bench.py (and its helper module tools/benchlib.py) import the oracle only inside cpu_baseline()
    for f in ("bench.py", os.path.join("tools", "benchlib.py")):
        src = open(os.path.join(REPO, f)).read()
        top_level = "\n".join(ln for ln in src.splitlines() if re.match(r"(from|import)\s", ln))
        assert "oracle" not in top_level, f
        inside = [ln for ln in src.splitlines() if re.match(r"\s+(from|import)\s.*oracle", ln)]
        assert len(inside) == (1 if f.endswith("benchlib.py") else 0), (f, inside)      # ... and exactly there
    lib = open(os.path.join(REPO, "tools", "benchlib.py")).read()
    assert lib.index("def cpu_baseline") < lib.index("from oracle import") < lib.index("def relaunch_command")


def test_reference_module_names_resolve_without_a_gpu():
    """The opt-in compat/gsplat package imports on a CPU-only box (the HIP library is only loaded
    on first use); in a subprocess, so that ``gsplat`` never enters this test process's module table."""
    import subprocess
    import sys
    from tests.conftest import REPO
    code = ("import importlib\n"
            "for name in ('gsplat.gau_io', 'gsplat.read_write_model', 'gsplat.utils', 'gsplat.pytorch_ssim',\n"
            "             'gsplat.gausplat_dataset', 'gsplat.gsmodel', 'gsplatcu'):\n"
            "    assert importlib.import_module(name) is not None\n"
            "from gsplat.gsmodel import GSModel, get_training_params\n"
            "from gsplat.gau_io import load_gs, save_gs, get_example_gs\n"
            "assert get_example_gs().shape == (4,)\n"
            "import gsplat.gsmodel as m; print(m.__file__)\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(REPO, "compat"), REPO]))
    r = subprocess.run([sys.executable, "-c", code], cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip().endswith(os.path.join("compat", "gsplat", "gsmodel.py"))


def test_missing_library_is_a_hard_error(monkeypatch, tmp_path):
    from easygaussiansplatting_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.EgsLibraryError):
        _lib.load()


def test_host_lr_schedule_matches_reference_samples():
    """density.expon_lr (host logic of gsmodel.py:180-183 / utils.py:7-44) against fixture G8's samples."""
    from tests.conftest import load_golden
    from easygaussiansplatting_amd.density import expon_lr
    g = load_golden("g8_densify.npz")
    got = [expon_lr(int(s), 1e-4 * 2.5, 1e-6 * 2.5, 3000, delay_mult=0.01) for s in g["lr_steps"]]
    np.testing.assert_allclose(got, g["lr_values"], rtol=1e-12)


def test_bench_flags_of_the_contract_and_of_this_build():
    """``python bench.py --help`` runs without a GPU and lists the driver's flags (--gpus / --steps / --warmup) and the
    ones DESIGN / profiles quote (--views-per-rank, --view-streams, --overlap-exchange, --no-ops, --no-ring8)."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--help"], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    for flag in ("--gpus", "--steps", "--warmup", "--views-per-rank", "--view-streams", "--overlap-exchange",
                 "--no-ops", "--no-ring8", "--cpu-sample", "--ramp-steps"):
        assert flag in out.stdout, flag


def test_bench_relaunches_itself_under_the_launcher_for_several_gpus():
    """``python bench.py --gpus N`` without WORLD_SIZE re-executes itself as the contract's command
    ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    --gpus N ...`` (one rank per GPU over RCCL); under a launcher, or with one GPU, it runs in place."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("egs_bench_module", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    argv = ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    cmd = bench.relaunch_command(4, {}, argv)
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1].isdigit()
    i = cmd.index(os.path.join(REPO, "bench.py"))
    assert cmd[i + 1:] == argv                                   # the flags travel unchanged
    assert bench.relaunch_command(4, {"MASTER_PORT": "29611"}, argv)[cmd.index("--master-port") + 1] == "29611"
    assert bench.relaunch_command(1, {}, ["--gpus", "1"]) is None
    assert bench.relaunch_command(4, {"WORLD_SIZE": "4", "RANK": "0"}, argv) is None     # already launched
    # and the launcher really accepts that command line (parse only: --help of torch.distributed.run)
    out = subprocess.run(cmd[:3] + ["--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "--nproc-per-node" in out.stdout.replace("_", "-")


def test_public_pair_memo_is_opt_in():
    """The content-validated ``splat`` -> ``splatB`` memo (module-global state in a drop-in) is OFF unless a caller asks
    for it, and bounded when on."""
    from easygaussiansplatting_amd import gsplatcu
    assert gsplatcu._memo_enabled is False and gsplatcu._splat_memo == {} and gsplatcu.MEMO_MAX == 8


def test_render_options_validate_and_default():
    from easygaussiansplatting_amd.function import GSFunction, RenderOptions
    o = RenderOptions()
    assert (o.mode, o.ops_use_records, o.accumulate, o.sh_sink, o.exchange) == ("fused", True, False, None, None)
    assert GSFunction.mode == "fused"
    with pytest.raises(ValueError):
        RenderOptions(mode="cuda")
    with pytest.raises(ValueError):
        RenderOptions(sh_sink=object(), exchange=object())


def test_variant_patches_apply():
    """the measurement probes are patches under tools/lab/variants, not #ifdefs in the product sources:
    they must keep applying to the current csrc/, and the sources must hold no probe knob"""
    import glob
    import subprocess
    patches = sorted(glob.glob(os.path.join(REPO, "tools", "lab", "variants", "*.patch")))
    assert len(patches) >= 2
    for p in patches:
        r = subprocess.run(["git", "apply", "--check", p], cwd=REPO, capture_output=True, text=True)
        assert r.returncode == 0, "%s no longer applies:\n%s" % (os.path.basename(p), r.stderr)
    for f in glob.glob(os.path.join(REPO, "easygaussiansplatting_amd", "csrc", "*.h*")):
        src = open(f).read()
        for knob in ("EGS_DRAW_PROBE_NOK", "EGS_PROBE_REDUCE", "EGS_DRAW_DUMMY_", "EGS_PLAN_STAMPS", "EGS_PROBE_HIT_BITS",
                     "WRONG"):
            assert knob not in src, "%s still holds %s" % (os.path.basename(f), knob)
