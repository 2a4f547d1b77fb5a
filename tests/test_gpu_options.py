"""Per-call configuration of the training path (``function.RenderOptions``): how a render is evaluated -- fused kernels
or the reference's seven-op structure, in-kernel gradient accumulation, the factored SH gradient -- travels with the
``GSFunction.apply`` call and its autograd node, not with class / module switches (two trainers, or a
trainer and a viewer, in one process shared ``GSFunction.mode`` and the ``fused.accumulate_in_kernel()`` block)."""
import numpy as np
import pytest
import torch

from easygaussiansplatting_amd import scene as S

pytestmark = pytest.mark.gpu


def _setup(n=2500, W=96, H=64, views=3):
    from easygaussiansplatting_amd.function import Camera
    sc = S.small_scene(n, W, H, 12, seed=31)
    cams = [Camera.from_scene(c) for c in S.ring_cameras(sc.cam, views, radius=5.0)]
    gts = [torch.from_numpy(np.clip(0.5 + 0.2 * S.normal(40 + i, 1, (3, H, W)), 0, 1).astype(np.float32)).cuda()
           for i in range(views)]
    return sc, cams, gts


def _trainer(sc, cams, gts, mode, **kw):
    from easygaussiansplatting_amd.trainer import Trainer
    return Trainer(sc, cams, gts, max_steps=50, fused_activations=False, mode=mode, view_streams=1, **kw)


def _state(tr):
    return {k: v.detach().cpu().numpy().copy() for k, v in tr.params.items()}


def test_two_trainers_with_different_modes_in_one_process():
    """Trainer(mode="ops") and Trainer(mode="fused") stepping ALTERNATELY in one process end exactly where each ends
    when it runs alone, and the process-wide defaults are never touched."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from easygaussiansplatting_amd.function import GSFunction
    sc, cams, gts = _setup()
    steps = [[0, 1], [2, 0], [1, 2]]
    alone = {}
    for mode in ("ops", "fused"):
        tr = _trainer(sc, cams, gts, mode)
        for v in steps:
            tr.step(v)
        alone[mode] = _state(tr)
    a, b = _trainer(sc, cams, gts, "ops"), _trainer(sc, cams, gts, "fused")
    for v in steps:                      # interleaved: a shared switch would make one of them run the other's path
        a.step(v)
        b.step(v)
        assert GSFunction.mode == "fused" and GSFunction.ops_use_records is True
    for tr, mode in ((a, "ops"), (b, "fused")):
        got = _state(tr)
        for k in got:
            ref = alone[mode][k]
            # (float atomics: the order of the additions differs from run to run)
            assert np.abs(got[k] - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), (mode, k)
    # ... and the two modes really are different evaluations of the same function: close, not identical
    d = max(np.abs(alone["ops"][k] - alone["fused"][k]).max() for k in alone["ops"])
    assert 0 < d < 1e-3


def test_render_options_are_per_call():
    """``GSFunction.apply(..., opts)``: the seven-op structure and the fused kernels through ONE process-wide default,
    in-kernel accumulation without the ``accumulate_in_kernel()`` block, a ``FactoredShGrad`` named by the call."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from easygaussiansplatting_amd import dist_views as DV
    from easygaussiansplatting_amd.function import GSFunction, RenderOptions
    sc, cams, _ = _setup()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    H, W = sc.cam.height, sc.cam.width
    dl = dev(S.normal(5, 3, (3, H, W)) / (3 * H * W))
    names = ("pws", "shs", "alphas", "scales", "rots")

    def leaves():
        p = dict(pws=dev(sc.pws), shs=dev(sc.shs), alphas=dev(sc.alphas).reshape(-1, 1), scales=dev(sc.scales),
                 rots=dev(sc.rots))
        for v in p.values():
            v.requires_grad_(True)
        return p

    def run(opts_of_view, fx=None):
        p = leaves()
        us = torch.zeros((sc.n, 2), device="cuda", requires_grad=True)
        if fx is not None:
            fx.begin_step(sc.n, "cuda")
        for i, c in enumerate(cams[:2]):
            o = opts_of_view(i)
            args = [p[k] for k in names] + [us, c] + ([o] if o is not None else [])
            img, _ = GSFunction.apply(*args)
            img.backward(dl)
        if fx is not None:
            fx.finish(p["pws"], p["shs"])
        return {k: p[k].grad.detach().cpu().numpy() for k in names}

    ref = run(lambda i: None)                                               # the defaults: fused, autograd accumulates
    for label, got in (
            ("ops", run(lambda i: RenderOptions(mode="ops"))),
            ("ops_public_pair", run(lambda i: RenderOptions(mode="ops", ops_use_records=False))),
            ("accumulate", run(lambda i: RenderOptions(accumulate=True))),
            ("mixed", run(lambda i: RenderOptions(mode="ops" if i == 0 else "fused", accumulate=(i == 1)))),
    ):
        for k in names:
            assert np.abs(got[k] - ref[k]).max() <= 3e-5 * np.abs(ref[k]).max(), (label, k)
    fx = DV.FactoredShGrad(2)
    got = run(lambda i: RenderOptions(accumulate=True, sh_sink=fx), fx)
    for k in names:
        assert np.abs(got[k] - ref[k]).max() <= 3e-5 * np.abs(ref[k]).max(), ("factored", k)
    assert GSFunction.mode == "fused"
    with pytest.raises(ValueError):
        RenderOptions(mode="triton")
