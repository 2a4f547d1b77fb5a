"""The SH gradient of a step in its factored form (``dist_views.FactoredShGrad``, ``EGS_BWD_FACTORED_SH``,
``egs_sh_grad_views``): eq (5) of backward.md (gsmodel.py:84-85) is an outer product per Gaussian and view, so a
view leaves dL/dcolour [N,3] and the 48-float rows are formed once per step.  Pinned here: the rows it forms are
the rows the chain-rule kernel writes (same basis function, same products), summed over the views of a step."""
import contextlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NAMES_RAW = ("pws", "low_shs", "high_shs", "alphas_raw", "scales_raw", "rots_raw")


def _setup(n, K, V, seed, H=96, W=160):
    from easygaussiansplatting_amd import scene as S
    from easygaussiansplatting_amd.function import Camera, GSFunction
    GSFunction.mode = "fused"
    sc = S.small_scene(n, W, H, K, seed=seed)
    sc.pws[:25, 2] = -9.0                                  # behind some of the cameras: rows without a gradient
    cams = [Camera.from_scene(c) for c in S.ring_cameras(sc.cam, max(V, 2), radius=5.0)][:V]
    dls = [torch.from_numpy(S.normal(9, v, (3, H, W)).astype(np.float32)).cuda() / (3 * H * W) for v in range(V)]
    return sc, cams, dls


def _leaves(sc, raw):
    from easygaussiansplatting_amd.trainer import raw_params_from_scene
    if raw:
        p = raw_params_from_scene(sc, "cuda")
        return [p[k] for k in NAMES_RAW]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    return [dev(a).requires_grad_(True) for a in (sc.pws, sc.shs, sc.alphas.reshape(-1, 1), sc.scales, sc.rots)]


def _sh(L, raw):
    return (L[1], L[2]) if raw else (L[1], None)


def _close(x, y, tol, what):
    scale = float(x.abs().max())
    assert scale > 0, what
    err = float((x - y).abs().max())
    assert err <= tol * scale, (what, err, scale)


@pytest.mark.parametrize("K", [48, 27, 12, 3])
@pytest.mark.parametrize("raw", [False, True])
def test_one_view_rows_are_the_chain_rule_kernels_rows(raw, K):
    """One view: the rows formed from dL/dcolour equal the rows k_preprocess_bwd writes -- to the last bit or two (the
    same products; only the atomics order of k_draw_bwd differs between two backward passes)."""
    from easygaussiansplatting_amd import dist_views as DV, fused
    from easygaussiansplatting_amd.function import GSFunction, GSRawFunction
    F = GSRawFunction if raw else GSFunction
    sc, cams, dls = _setup(4003, K, 1, 41)
    A = _leaves(sc, raw)
    us = torch.zeros((sc.n, 2), device="cuda", requires_grad=True)
    F.apply(*A, us, cams[0])[0].backward(dls[0])
    B = _leaves(sc, raw)
    us_b = torch.zeros((sc.n, 2), device="cuda", requires_grad=True)
    fx = DV.FactoredShGrad(views=1)
    with fx.attach():
        F.apply(*B, us_b, cams[0])[0].backward(dls[0])
    sh, high = _sh(B, raw)
    assert sh.grad is None and (high is None or high.grad is None)      # autograd got None for the SH tensors
    others = [t for t in B if t is not sh and t is not high]
    flat = fused.flat_grad_buffer(others)                               # the other four still tile ONE buffer
    assert flat is not None and 11 * sc.n <= flat.numel() < 11 * sc.n + 16
    fx.finish(B[0], sh, high)
    torch.cuda.synchronize()
    for k, (a, b) in enumerate(zip(A, B)):
        if a.numel() == 0:
            continue
        assert b.grad is not None and b.grad.shape == a.grad.shape, k
        _close(a.grad, b.grad, 2e-5, k)
    assert not B[1].grad[:25].any()
    _close(us.grad, us_b.grad, 2e-5, "us")


@pytest.mark.parametrize("lanes", [1, 3])
@pytest.mark.parametrize("raw", [False, True])
def test_views_of_a_step(raw, lanes):
    """Five views per step, one after the other or dealt to three streams (``ViewStreams``), with the in-kernel
    accumulation of the other four tensors: the step's gradients equal the plain accumulation of five backward
    passes; a second step through the same objects starts clean; an existing ``.grad`` is added to."""
    from easygaussiansplatting_amd import dist_views as DV, fused
    from easygaussiansplatting_amd.function import GSFunction, GSRawFunction
    F = GSRawFunction if raw else GSFunction
    V = 5
    sc, cams, dls = _setup(6001, 48, V, 43)
    A = _leaves(sc, raw)
    us = torch.zeros((sc.n, 2), device="cuda", requires_grad=True)
    for cam, dl in zip(cams, dls):
        F.apply(*A, us, cam)[0].backward(dl)
    torch.cuda.synchronize()
    B = _leaves(sc, raw)
    sh, high = _sh(B, raw)
    vs = DV.ViewStreams(B, lanes) if lanes > 1 else None
    fx = DV.FactoredShGrad(views=V)
    for rep in range(3):
        if rep < 2:
            for t in B:
                t.grad = None
        uss = [torch.zeros((sc.n, 2), device="cuda", requires_grad=True) for _ in range(lanes)]
        if vs is not None:
            vs.begin()
        with fx.attach(), fused.accumulate_in_kernel():
            for i, (cam, dl) in enumerate(zip(cams, dls)):
                with (vs.lane(i) if vs is not None else contextlib.nullcontext(B)) as lv:
                    F.apply(*lv, uss[i % lanes], cam)[0].backward(dl)
        if vs is not None:
            vs.finish()
        fx.finish(B[0], sh, high)
        torch.cuda.synchronize()
        mult = 2.0 if rep == 2 else 1.0          # third round: nothing was cleared, everything is added once more
        for k, (a, b) in enumerate(zip(A, B)):
            if a.numel():
                _close(a.grad * mult, b.grad, 4e-5, (rep, k))
        _close(us.grad, sum(u.grad for u in uss), 4e-5, "us")


def test_autograd_grad_and_unattached_passes_get_their_rows():
    """``torch.autograd.grad`` inside ``attach()`` returns the SH rows as always (nothing may be left in a sink the
    caller never finishes); a ``.backward()`` outside ``attach()`` is untouched; a pass beyond ``views`` raises."""
    from easygaussiansplatting_amd import dist_views as DV
    from easygaussiansplatting_amd.function import GSFunction
    sc, cams, dls = _setup(3000, 48, 2, 47)
    A = _leaves(sc, False)
    us = torch.zeros((sc.n, 2), device="cuda", requires_grad=True)
    GSFunction.apply(*A, us, cams[0])[0].backward(dls[0])
    B = _leaves(sc, False)
    fx = DV.FactoredShGrad(views=1)
    with fx.attach():
        img = GSFunction.apply(*B, torch.zeros((sc.n, 2), device="cuda", requires_grad=True), cams[0])[0]
        g = torch.autograd.grad(img, B, dls[0])
    assert all(t.grad is None for t in B) and fx._next == 0
    for a, b in zip(A, g):
        _close(a.grad, b, 2e-5, "autograd.grad")
    with fx.attach():
        GSFunction.apply(*B, us, cams[0])[0].backward(dls[0])
        with pytest.raises(RuntimeError, match="views=1"):
            GSFunction.apply(*B, us, cams[1])[0].backward(dls[1])


def test_sh_grad_views_c_abi_against_the_oracle_basis():
    """``egs_sh_grad_views`` alone, through the C ABI: scale * sum_v dcolour_v (x) basis(pw - twc_v) with the basis
    of the pinned oracle (``sh_basis`` == ``sh2color``'s dcolor/dsh, backward_cpu.py:278-385), zero rows skipped,
    accumulation into existing values, all four SH widths, both layouts."""
    import oracle.gs_oracle as O
    from easygaussiansplatting_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    n, V = 1531, 4
    pws = rng.normal(size=(n, 3)).astype(np.float32) * 3
    twcs = rng.normal(size=(V, 3)).astype(np.float32) * 6
    dcol = rng.normal(size=(V, n, 3)).astype(np.float32)
    dcol[1, ::3] = 0.0
    dcol[2] = 0.0                                           # a view nobody rendered
    stride = (3 * n + 6) // 4 * 4
    rows = np.full((V, stride), np.nan, np.float32)         # padding is never read
    rows[:, :3 * n] = dcol.reshape(V, -1)
    rows[:, 3 * n:3 * n + 3] = twcs
    d_rows, d_pws = torch.from_numpy(rows).cuda(), torch.from_numpy(pws).cuda()
    st = torch.cuda.current_stream().cuda_stream
    for K in (3, 12, 27, 48):
        want = np.zeros((n, K), np.float64)
        for v in range(V):
            d = pws.astype(np.float64) - twcs[v].astype(np.float64)
            basis = O.sh_basis(d / np.sqrt((d * d).sum(1))[:, None], K // 3)[:, :K // 3]     # == dcolour/dsh
            want += (dcol[v].astype(np.float64)[:, None, :] * basis[:, :, None]).reshape(n, K)  # sh[i, 3 c + rgb]
        want *= 0.25
        out = torch.full((n, K), 7.0, device="cuda")
        _lib.check(lib.egs_sh_grad_views(n, K, V, d_pws.data_ptr(), d_rows.data_ptr(), stride, 0.25, out.data_ptr(),
                                         None, 0, st))
        got = out.cpu().numpy()
        assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max(), K
        _lib.check(lib.egs_sh_grad_views(n, K, V, d_pws.data_ptr(), d_rows.data_ptr(), stride, 0.25, out.data_ptr(),
                                         None, 1, st))
        assert np.abs(out.cpu().numpy() - 2 * want).max() <= 4e-6 * np.abs(want).max(), K
        if K > 3:
            low, high = torch.zeros((n, 3), device="cuda"), torch.zeros((n, K - 3), device="cuda")
            _lib.check(lib.egs_sh_grad_views(n, K, V, d_pws.data_ptr(), d_rows.data_ptr(), stride, 0.25,
                                             low.data_ptr(), high.data_ptr(), 0, st))
            assert torch.equal(torch.cat((low, high), 1).cpu(), torch.from_numpy(got)), K
    # no views: zeros (or nothing added)
    out = torch.full((n, 48), 7.0, device="cuda")
    _lib.check(lib.egs_sh_grad_views(n, 48, 0, d_pws.data_ptr(), None, stride, 1.0, out.data_ptr(), None, 0, st))
    assert not out.any()


@pytest.mark.parametrize("K", [48, 12, 3])
@pytest.mark.parametrize("raw", [False, True])
def test_fused_adam_from_factored_rows_equals_rows_then_adam(raw, K):
    """``FusedAdam.step(factored_sh=...)`` (``egs_adam_sh_factored``: every Gaussian's SH gradient row formed in LDS
    and consumed there) == ``egs_sh_grad_views`` into ``.grad`` followed by the ordinary step, on the same rows:
    parameters and both moments, three steps, two views, a row count that is not a multiple of 64."""
    from easygaussiansplatting_amd import _lib, dist_views as DV
    from easygaussiansplatting_amd.optim import FusedAdam
    lib = _lib.load()
    torch.manual_seed(3)
    n, V = 4037, 2
    stride = DV.FactoredShGrad.row_stride(n)
    pws = (torch.randn(n, 3, device="cuda") * 3).contiguous()

    def tensors():
        g = torch.Generator(device="cuda").manual_seed(11)
        if raw and K > 3:
            return [torch.randn(n, 3, device="cuda", generator=g).requires_grad_(True),
                    torch.randn(n, K - 3, device="cuda", generator=g).requires_grad_(True)]
        return [torch.randn(n, K, device="cuda", generator=g).requires_grad_(True)]

    def groups(ts):
        return [{"params": [t], "lr": lr, "name": nm} for t, lr, nm in zip(ts, (1e-3, 5e-5), ("low_shs", "high_shs"))]
    A, B = tensors(), tensors()
    oa, ob = FusedAdam(groups(A), eps=1e-15), FusedAdam(groups(B), eps=1e-15)
    st = torch.cuda.current_stream().cuda_stream
    for step in range(3):
        rows = torch.randn(V, stride, device="cuda") * 1e-3
        rows[1, 0:3 * n:7] = 0.0
        rows[:, 3 * n:3 * n + 3] = torch.randn(V, 3, device="cuda") * 6
        grads = [torch.empty_like(t) for t in A]
        _lib.check(lib.egs_sh_grad_views(n, K, V, pws.data_ptr(), rows.data_ptr(), stride, 0.5, grads[0].data_ptr(),
                                         grads[1].data_ptr() if len(grads) > 1 else None, 0, st))
        for t, g in zip(A, grads):
            t.grad = g
        oa.step()
        for t in B:
            t.grad = None
        ob.step(factored_sh=(rows, 0.5, pws, B[0], B[1] if len(B) > 1 else None))
        torch.cuda.synchronize()
        for a, b in zip(A, B):
            assert torch.equal(a.detach(), b.detach()), (step, float((a - b).abs().max()))
            for key in ("exp_avg", "exp_avg_sq"):
                assert torch.equal(oa.state[a][key], ob.state[b][key]), (step, key)
            assert oa.state[a]["step"] == ob.state[b]["step"] == step + 1
    assert float((A[0].detach() - tensors()[0].detach()).abs().max()) > 1e-4      # the parameters did move


@pytest.mark.parametrize("views", [[0], [0, 1, 2]])
def test_trainer_steps_agree_across_the_sh_gradient_forms(views):
    """``Trainer.step`` three ways -- the SH rows written per view (``factored_sh=False``), the factored form expanded
    into ``.grad`` for ``torch.optim.Adam`` (``FactoredShGrad.finish``), the factored form consumed by ``FusedAdam``
    (``egs_adam_sh_factored``) -- takes the same three optimizer steps: losses, parameters and the densification
    statistics agree to the order of the float sums."""
    from easygaussiansplatting_amd import scene as S
    from easygaussiansplatting_amd.function import Camera, render
    from easygaussiansplatting_amd.trainer import Trainer
    sc = S.small_scene(12000, 192, 128, 48, seed=8)
    cams = [Camera.from_scene(c) for c in S.ring_cameras(sc.cam, 3, radius=5.0)]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    with torch.no_grad():
        gts = [render(dev(sc.pws), dev(sc.shs), dev(sc.alphas), dev(sc.scales), dev(sc.rots), c)[0] for c in cams]
    outs = []
    for fused_adam, factored in ((True, False), (False, True), (True, True)):
        start = S.small_scene(12000, 192, 128, 48, seed=8)
        start.shs[:, :3] += 0.4
        tr = Trainer(start, cams, gts, max_steps=100, scene_size=4.0, fused_adam=fused_adam, factored_sh=factored)
        losses = [tr.step(views) for _ in range(3)]
        outs.append((losses, {k: v.detach().cpu().numpy() for k, v in tr.params.items()},
                     tr.grad_accum.cpu().numpy(), tr.vis_count.cpu().numpy()))
    for o in outs[1:]:
        np.testing.assert_allclose(o[0], outs[0][0], rtol=2e-5)
        assert (o[3] == outs[0][3]).all()
        np.testing.assert_allclose(o[2], outs[0][2], rtol=2e-4, atol=1e-9)
        for k in o[1]:
            a, b = outs[0][1][k], o[1][k]
            # three Adam steps of size lr (the update is normalised: a strict check of the gradient's direction)
            assert np.abs(a - b).max() < 2e-4 * max(1e-3, np.abs(a).max()), k
