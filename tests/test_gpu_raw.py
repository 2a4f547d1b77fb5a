"""GSRawFunction (activations of gsplat/utils.py:121-150 inside the fused kernels) against the
reference's structure: torch activations + GSFunction (gsmodel.py:198-210)."""
import numpy as np
import pytest
import torch

from tests.gradcheck import assert_grad_close

pytestmark = pytest.mark.gpu


def _raw_params(n, K, seed, w=160, h=96):
    from easygaussiansplatting_amd import scene as S
    from easygaussiansplatting_amd.function import Camera
    sc = S.small_scene(n, w, h, K, seed=seed)
    g = torch.Generator().manual_seed(seed)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32))
    alphas = t(sc.alphas).clamp(1e-3, 1 - 1e-3)
    p = {"pws": t(sc.pws), "low_shs": t(sc.shs[:, :3]).contiguous(), "high_shs": t(sc.shs[:, 3:]).contiguous(),
         "alphas_raw": torch.log(alphas / (1 - alphas)).reshape(-1, 1),
         "scales_raw": torch.log(t(sc.scales)),
         "rots_raw": t(sc.rots) * (0.2 + 3 * torch.rand(n, 1, generator=g))}      # un-normalised on purpose
    p["pws"][: n // 50, 2] = -50.0                                                # some behind the camera
    return {k: v.cuda().contiguous() for k, v in p.items()}, Camera.from_scene(sc.cam)


@pytest.mark.parametrize("K", [48, 12, 3])
def test_raw_function_matches_torch_activations(K):
    from easygaussiansplatting_amd import gsplatcu as gsc
    from easygaussiansplatting_amd.function import GSFunction, GSRawFunction
    gsc.set_policy("gsplatcu")
    n = 5000
    base, cam = _raw_params(n, K, 11)
    # (seeded: with the process-wide generator the image gradient -- and with it which Gaussians sit on a threshold --
    # depended on which tests had drawn random numbers before this one)
    dl = torch.randn(3, cam.height, cam.width, device="cuda",
                     generator=torch.Generator(device="cuda").manual_seed(100 + K)) / (3 * cam.height * cam.width)
    names = ("pws", "low_shs", "high_shs", "alphas_raw", "scales_raw", "rots_raw")

    def run(raw):
        p = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        us = torch.zeros(n, 2, device="cuda", requires_grad=True)
        if raw:
            img, mask = GSRawFunction.apply(*[p[k] for k in names], us, cam)
        else:
            img, mask = GSFunction.apply(p["pws"], torch.cat((p["low_shs"], p["high_shs"]), dim=1),
                                         torch.sigmoid(p["alphas_raw"]), torch.exp(p["scales_raw"]),
                                         torch.nn.functional.normalize(p["rots_raw"]), us, cam)
        img.backward(dl)
        grads = {k: (p[k].grad if p[k].grad is not None else torch.zeros_like(p[k])) for k in names}
        grads["us"] = us.grad
        return img.detach(), mask, grads

    img_a, mask_a, ga = run(False)
    img_b, mask_b, gb = run(True)
    assert torch.equal(mask_a, mask_b) and int(mask_a.sum()) < n
    assert float((img_a - img_b).abs().max()) < 2e-5
    for k in ga:
        a, b = ga[k].cpu().numpy(), gb[k].cpu().numpy()
        assert a.shape == b.shape, k
        if a.size == 0:          # K == 3: high_shs has no columns
            continue
        # relative to the gradient's own magnitude (tests/gradcheck.py); the two paths feed the kernels inputs that
        # differ in the last bit (sigmoid(logit), exp(log)), so a few threshold-flip Gaussians exist: counted
        assert_grad_close(b, a, "raw_vs_fused[%d]:%s" % (K, k), outliers=4)


def test_raw_function_trainer_equivalence_20k():
    """20 k Gaussians, two views: three Trainer steps with the activations inside the kernels or in torch move
    the parameters identically (Adam normalises the gradient, so this is a strict check of its direction).
    (The 1 M / 1080p comparison of the two paths is tests/test_gpu_parity.py::test_full_size_fused_and_raw_paths.)"""
    from easygaussiansplatting_amd import scene as S
    from easygaussiansplatting_amd.function import Camera, render
    from easygaussiansplatting_amd.trainer import Trainer
    sc = S.small_scene(20000, 256, 144, 48, seed=2)
    cams = [Camera.from_scene(c) for c in S.ring_cameras(sc.cam, 2, radius=5.0)]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    with torch.no_grad():
        gts = [render(dev(sc.pws), dev(sc.shs), dev(sc.alphas), dev(sc.scales), dev(sc.rots), c)[0] for c in cams]
    outs = []
    for fused_act in (False, True):
        start = S.small_scene(20000, 256, 144, 48, seed=2)
        start.shs[:, :3] += 0.5
        tr = Trainer(start, cams, gts, max_steps=100, scene_size=4.0, fused_activations=fused_act)
        losses = [tr.step([0, 1]) for _ in range(3)]
        outs.append((losses, {k: v.detach().cpu().numpy() for k, v in tr.params.items()}))
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=2e-4)
    for k in outs[0][1]:
        a, b = outs[0][1][k], outs[1][1][k]
        # three Adam steps of size lr: parameters agree to a small fraction of the distance moved
        assert np.abs(a - b).max() < 2e-3 * max(1e-3, np.abs(a).max()), k


@pytest.mark.parametrize("n", [4001, 5000])
def test_parameter_gradients_share_one_buffer(n):
    """The fused backward hands out the parameter gradients as 16-B aligned slices of one allocation, and
    autograd adopts them as .grad without copying: a data-parallel caller all-reduces ONE tensor."""
    from easygaussiansplatting_amd import fused
    from easygaussiansplatting_amd.function import GSFunction, GSRawFunction
    base, cam = _raw_params(n, 48, 5)
    dl = torch.randn(3, cam.height, cam.width, device="cuda",
                     generator=torch.Generator(device="cuda").manual_seed(n)) / (3 * cam.height * cam.width)
    p = {k: v.clone().requires_grad_(True) for k, v in base.items()}
    us = torch.zeros(n, 2, device="cuda", requires_grad=True)
    names = ("pws", "low_shs", "high_shs", "alphas_raw", "scales_raw", "rots_raw")
    img, _ = GSRawFunction.apply(*[p[k] for k in names], us, cam)
    img.backward(dl)
    flat = fused.flat_grad_buffer([p[k] for k in names])
    assert flat is not None and flat.numel() >= 59 * n and flat.numel() < 59 * n + 24
    for k in names:
        assert p[k].grad.data_ptr() % 16 == 0
        assert p[k].grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr()
    before = p["high_shs"].grad.clone()
    flat.mul_(2.0)                                       # what an all-reduce + scale does: in place, all at once
    assert torch.equal(p["high_shs"].grad, before * 2)
    # the activated-parameter function too
    q = [base["pws"].clone(), torch.cat((base["low_shs"], base["high_shs"]), 1), torch.sigmoid(base["alphas_raw"]),
         torch.exp(base["scales_raw"]), torch.nn.functional.normalize(base["rots_raw"])]
    q = [t.contiguous().requires_grad_(True) for t in q]
    img, _ = GSFunction.apply(*q, torch.zeros(n, 2, device="cuda", requires_grad=True), cam)
    img.backward(dl)
    assert fused.flat_grad_buffer(q) is not None
    assert fused.flat_grad_buffer(q[:3]) is None         # does not tile the buffer


@pytest.mark.parametrize("raw", [False, True])
def test_accumulate_in_kernel_equals_autograd_accumulation(raw):
    """Several views per step: inside ``fused.accumulate_in_kernel()`` the chain-rule kernel adds a view's parameter
    gradients to the leaves' ``.grad`` itself (EGS_BWD_ACCUMULATE) and autograd is handed None -- same sums as
    autograd's own accumulation of per-view tensors, the ``.grad`` buffers stay the ONE flat allocation."""
    import numpy as np
    from easygaussiansplatting_amd import fused, scene as S
    from easygaussiansplatting_amd.function import Camera, GSFunction, GSRawFunction
    from easygaussiansplatting_amd.trainer import raw_params_from_scene
    GSFunction.mode = "fused"
    sc = S.small_scene(5000, 160, 96, 48, seed=31)
    sc.pws[:30, 2] = -9.0                                  # some culled rows: they must keep what they hold
    cams = [Camera.from_scene(c) for c in S.ring_cameras(sc.cam, 3, radius=5.0)]
    dls = [torch.from_numpy(S.normal(6, v, (3, 96, 160)).astype(np.float32)).cuda() / (3 * 96 * 160) for v in range(3)]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()

    def leaves():
        if raw:
            p = raw_params_from_scene(sc, "cuda")
            return [p[k] for k in ("pws", "low_shs", "high_shs", "alphas_raw", "scales_raw", "rots_raw")]
        return [dev(a).requires_grad_(True) for a in (sc.pws, sc.shs, sc.alphas.reshape(-1, 1), sc.scales, sc.rots)]

    def run(in_kernel):
        L = leaves()
        us = torch.zeros((sc.n, 2), device="cuda", requires_grad=True)
        import contextlib
        with (fused.accumulate_in_kernel() if in_kernel else contextlib.nullcontext()):
            for cam, dl in zip(cams, dls):
                img, _ = (GSRawFunction if raw else GSFunction).apply(*L, us, cam)
                img.backward(dl)
        torch.cuda.synchronize()
        assert fused.flat_grad_buffer(L) is not None
        return [t.grad.clone() for t in L] + [us.grad.clone()]
    a, b = run(False), run(True)
    for x, y in zip(a, b):
        scale = float(x.abs().max())
        assert scale > 0 and float((x - y).abs().max()) <= 2e-5 * scale     # atomics order of k_draw_bwd only
    assert not a[0][:30].any() and not b[0][:30].any()


@pytest.mark.parametrize("n_streams", [2, 3])
@pytest.mark.parametrize("raw", [False, True])
def test_view_streams_equal_sequential_views(raw, n_streams):
    """``dist_views.ViewStreams``: the views of a step dealt to 2 / 3 HIP streams (each with its own leaf aliases and
    gradient accumulator, added by ``finish()``) give the gradients of the same views rendered one after another --
    up to the order of the float sums -- and every view's image bit for bit; twice in a row through the SAME object
    (the second step must not see the first one's accumulators)."""
    import numpy as np
    from easygaussiansplatting_amd import dist_views as DV, fused, scene as S
    from easygaussiansplatting_amd.function import Camera, GSFunction, GSRawFunction
    from easygaussiansplatting_amd.trainer import raw_params_from_scene
    GSFunction.mode = "fused"
    H, W, V = 96, 160, 5
    sc = S.small_scene(6000, W, H, 48, seed=37)
    sc.pws[:20, 2] = -9.0                                  # (culled in some of the views)
    cams = [Camera.from_scene(c) for c in S.ring_cameras(sc.cam, V, radius=5.0)]
    dls = [torch.from_numpy(S.normal(8, v, (3, H, W)).astype(np.float32)).cuda() / (3 * H * W) for v in range(V)]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    F = GSRawFunction if raw else GSFunction

    def leaves():
        if raw:
            p = raw_params_from_scene(sc, "cuda")
            return [p[k] for k in ("pws", "low_shs", "high_shs", "alphas_raw", "scales_raw", "rots_raw")]
        return [dev(a).requires_grad_(True) for a in (sc.pws, sc.shs, sc.alphas.reshape(-1, 1), sc.scales, sc.rots)]

    L = leaves()
    us = torch.zeros((sc.n, 2), device="cuda", requires_grad=True)
    ref_imgs = []
    with fused.accumulate_in_kernel():
        for cam, dl in zip(cams, dls):
            img, _ = F.apply(*L, us, cam)
            img.backward(dl)
            ref_imgs.append(img.detach().clone())
    torch.cuda.synchronize()
    ref = [t.grad.clone() for t in L]

    M = leaves()
    vs = DV.ViewStreams(M, n_streams)
    for rep in range(2):
        for t in M:
            t.grad = None
        uss = [torch.zeros((sc.n, 2), device="cuda", requires_grad=True) for _ in range(n_streams)]
        imgs = [None] * V
        vs.begin()
        with fused.accumulate_in_kernel():
            for i, (cam, dl) in enumerate(zip(cams, dls)):
                with vs.lane(i) as lv:
                    img, _ = F.apply(*lv, uss[vs.lane_index(i)], cam)
                    img.backward(dl)
                    imgs[i] = img.detach()
        vs.finish()
        torch.cuda.synchronize()
        for a, b in zip(ref_imgs, imgs):
            assert torch.equal(a, b)
        for x, t in zip(ref, M):
            scale = float(x.abs().max())
            assert scale > 0 and float((x - t.grad).abs().max()) <= 3e-5 * scale, (rep, float((x - t.grad).abs().max()), scale)
        dus = sum(u.grad for u in uss if u.grad is not None)
        assert float((dus - us.grad).abs().max()) <= 3e-5 * float(us.grad.abs().max())


@pytest.mark.parametrize("fused_act", [False, True])
def test_trainer_view_streams_same_steps_as_one_stream(fused_act):
    """``Trainer.step`` with a rank's views on three streams (one gradient accumulator and one set of densification
    statistics per stream) takes the steps of the one-stream trainer: losses, statistics and parameters after three
    optimizer steps over five views agree to the order of the float sums; densification then re-allocates the
    parameters and the next step re-aliases the lanes."""
    import numpy as np
    from easygaussiansplatting_amd import scene as S
    from easygaussiansplatting_amd.function import Camera, render
    from easygaussiansplatting_amd.trainer import Trainer
    sc = S.small_scene(20000, 256, 144, 48, seed=4)
    cams = [Camera.from_scene(c) for c in S.ring_cameras(sc.cam, 5, radius=5.0)]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    with torch.no_grad():
        gts = [render(dev(sc.pws), dev(sc.shs), dev(sc.alphas), dev(sc.scales), dev(sc.rots), c)[0] for c in cams]
    outs = []
    for lanes in (1, 3):
        start = S.small_scene(20000, 256, 144, 48, seed=4)
        start.shs[:, :3] += 0.5
        tr = Trainer(start, cams, gts, max_steps=100, scene_size=4.0, fused_activations=fused_act, view_streams=lanes)
        losses = [tr.step([0, 1, 2, 3, 4]) for _ in range(3)]
        stats = (tr.grad_accum.cpu().numpy(), tr.vis_count.cpu().numpy())
        params = {k: v.detach().cpu().numpy() for k, v in tr.params.items()}
        tr.density.grad_threshold = float(np.percentile(stats[0] / np.maximum(stats[1], 1), 80))
        n0 = tr.params["pws"].shape[0]
        tr.densify()
        assert tr.params["pws"].shape[0] != n0
        after = tr.step([4, 3, 2, 1, 0])
        assert np.isfinite(after) and all(torch.isfinite(v).all() for v in tr.params.values())
        outs.append((losses, stats, params, tr.params["pws"].shape[0]))
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=2e-5)
    assert (outs[0][1][1] == outs[1][1][1]).all()                     # visibility counts: integers, exact
    np.testing.assert_allclose(outs[0][1][0], outs[1][1][0], rtol=2e-4, atol=1e-9)
    for k in outs[0][2]:
        a, b = outs[0][2][k], outs[1][2][k]
        assert np.abs(a - b).max() < 2e-4 * max(1e-3, np.abs(a).max()), k
    assert outs[0][3] == outs[1][3]                                    # the same Gaussians were cloned / split / pruned
