"""Fused forward with the draw stage enqueued ahead of the read-back of the patch count
(egs_splat_draw_rec_dev): same results as the synchronous path, safe on capacity / hint overflow."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(n, w, h, seed):
    from easygaussiansplatting_amd import scene as S
    from easygaussiansplatting_amd.function import Camera
    sc = S.small_scene(n, w, h, 12, seed=seed)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    return (dev(sc.pws), dev(sc.shs), dev(sc.alphas), dev(sc.scales), dev(sc.rots)), Camera.from_scene(sc.cam)


def _run(args, cam):
    from easygaussiansplatting_amd import fused
    img, mask, st = fused.forward(*args, cam)
    torch.cuda.synchronize()
    return [x.cpu().numpy() for x in (img, mask, st.ranges, st.gsid, st.contrib, st.final_tau, st.depths)]


@pytest.mark.parametrize("policy", ["gsplatcu", "forward_cpu"])
def test_ahead_equals_exact_and_survives_overflow(policy):
    from easygaussiansplatting_amd import fused
    from easygaussiansplatting_amd import gsplatcu as gsc
    gsc.set_policy(policy)
    try:
        args, cam = _scene(6000, 200, 120, 3)
        key = (6000, 200, 120)
        cap = fused._ctx(torch.device("cuda", 0)).capacity
        cap.pop(key, None)
        ref = _run(args, cam)                               # no capacity yet: synchronous read-back
        assert cap[key] > ref[3].shape[0] > 1000
        got = _run(args, cam)                               # enqueued ahead
        for a, b in zip(ref, got):
            np.testing.assert_array_equal(a, b)
        cap[key] = 64                                       # far too small: nothing out of bounds, draw redone
        got = _run(args, cam)
        for a, b in zip(ref, got):
            np.testing.assert_array_equal(a, b)
        assert cap[key] > ref[3].shape[0]
        cap[key] = ref[3].shape[0] - 7                      # just too small
        got = _run(args, cam)
        for a, b in zip(ref, got):
            np.testing.assert_array_equal(a, b)
        gsc._set_key_bits(0, key, 1)                        # stale hint: detected from the returned max key
        got = _run(args, cam)
        for a, b in zip(ref, got):
            np.testing.assert_array_equal(a, b)
        assert 8 <= gsc._get_key_bits(0, key) <= 32
        s = torch.cuda.Stream()                             # and on a non-default stream
        with torch.cuda.stream(s):
            got = _run(args, cam)
        for a, b in zip(ref, got):
            np.testing.assert_array_equal(a, b)
        far = (args[0] + torch.tensor([0.0, 0.0, -1000.0], device="cuda"),) + args[1:]   # nothing visible
        out = _run(far, cam)
        assert out[3].shape[0] == 0 and not out[0].any() and not out[5].any()
    finally:
        gsc.set_policy("gsplatcu")


def test_deferred_validation_reports_incomplete_renders():
    """Inside ``fused.deferred()`` nothing waits for the patch count; ``commit()`` names the renders whose
    capacity / depth-key hint was exceeded (and only those), and a re-render is exact."""
    from easygaussiansplatting_amd import fused
    from easygaussiansplatting_amd import gsplatcu as gsc
    args, cam = _scene(6000, 200, 120, 5)
    key = (6000, 200, 120)
    cap = fused._ctx(torch.device("cuda", 0)).capacity
    cap.pop(key, None)
    ref = _run(args, cam)
    P = ref[3].shape[0]
    with fused.deferred() as d:
        outs = [fused.forward(*args, cam) for _ in range(3)]
        assert all(o[2].ticket is not None or o[2]._patches is not None for o in outs)
        assert d.commit() == []                              # all complete
        for img, mask, st in outs:
            assert st.ticket is None and st.patch_count() == P and st.gsid.shape[0] == P
            np.testing.assert_array_equal(img.cpu().numpy(), ref[0])
            np.testing.assert_array_equal(st.gsid.cpu().numpy(), ref[3])
        cap[key] = P - 5                                     # the next render cannot hold its patch list
        img_bad, _, st_bad = fused.forward(*args, cam)
        bad = d.commit()
        assert len(bad) == 1 and bad[0] is st_bad and st_bad.patch_count() == P
        assert cap[key] > P                                  # learnt: the re-render fits
        img2, _, st2 = fused.forward(*args, cam)
        assert d.commit() == []
        np.testing.assert_array_equal(img2.cpu().numpy(), ref[0])
        gsc._set_key_bits(0, key, 2)                         # stale depth-key hint: same report
        _, _, st3 = fused.forward(*args, cam)
        bad = d.commit()
        assert len(bad) == 1 and bad[0] is st3
        img4, _, st4 = fused.forward(*args, cam)
        assert d.commit() == []
        np.testing.assert_array_equal(img4.cpu().numpy(), ref[0])
    # an uncollected failure must not pass silently
    cap[key] = 64
    with pytest.raises(RuntimeError, match="incomplete"):
        with fused.deferred():
            fused.forward(*args, cam)
    torch.cuda.synchronize()


def test_deferred_backward_equals_immediate():
    """Gradients of a render validated at commit() (capacity-sized patch list) == the immediate path."""
    from easygaussiansplatting_amd import fused
    from easygaussiansplatting_amd.function import GSFunction
    args, cam = _scene(5000, 160, 96, 9)
    dl = torch.randn((3, 96, 160), device="cuda", generator=torch.Generator(device="cuda").manual_seed(21)) / (3 * 96 * 160)

    def grads(deferred):
        ps = [a.clone().requires_grad_(True) for a in args]
        ps[2] = ps[2].detach().reshape(-1, 1).clone().requires_grad_(True)
        us = torch.zeros((5000, 2), device="cuda", requires_grad=True)
        if deferred:
            with fused.deferred() as d:
                img, _ = GSFunction.apply(ps[0], ps[1], ps[2], ps[3], ps[4], us, cam)
                img.backward(dl)
                assert d.commit() == []
        else:
            img, _ = GSFunction.apply(ps[0], ps[1], ps[2], ps[3], ps[4], us, cam)
            img.backward(dl)
        torch.cuda.synchronize()
        return [img.detach().cpu().numpy()] + [p.grad.cpu().numpy() for p in ps] + [us.grad.cpu().numpy()]
    a = grads(False); a = grads(False)      # second call: enqueue-ahead, validated at once
    b = grads(True)
    np.testing.assert_array_equal(a[0], b[0])
    for x, y in zip(a[1:], b[1:]):          # atomics: order of float additions differs from run to run
        assert np.abs(x).max() > 0
        np.testing.assert_allclose(x, y, rtol=0, atol=2e-5 * np.abs(x).max())


def test_second_backward_with_retain_graph():
    """``retain_graph=True`` keeps the saved state alive: a second backward gives the same gradients."""
    from easygaussiansplatting_amd.function import GSFunction
    args, cam = _scene(3000, 128, 96, 11)
    ps = [a.clone().requires_grad_(True) for a in args]
    ps[2] = ps[2].detach().reshape(-1, 1).clone().requires_grad_(True)
    us = torch.zeros((3000, 2), device="cuda", requires_grad=True)
    img, _ = GSFunction.apply(ps[0], ps[1], ps[2], ps[3], ps[4], us, cam)
    dl = torch.randn(img.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(22)) / img.numel()
    img.backward(dl, retain_graph=True)
    g1 = ps[0].grad.clone()
    ps[0].grad = None
    img.backward(dl)
    torch.cuda.synchronize()
    np.testing.assert_allclose(ps[0].grad.cpu().numpy(), g1.cpu().numpy(), rtol=0,
                               atol=2e-5 * float(g1.abs().max()))


def test_tile_dispatch_order_is_a_sorted_permutation():
    """``k_tile_order``: the dispatch order the draw kernels use is a permutation of the tiles -- longest list first
    on a camera's first render, most measured work first on its second one, then kept as it stands (no order
    kernel) and refreshed from the latest work every ``ORDER_REFRESH``-th render.  A camera keeps ONE
    ``[order | work]`` buffer (``state.order``); the image never depends on the order."""
    from easygaussiansplatting_amd import fused
    args, cam = _scene(20000, 640, 368, 4)
    T = (640 // 16) * (368 // 16)
    snaps, imgs = [], []
    keep_seg, fused.SEGMENTS = fused.SEGMENTS, "0"           # (the segment path plans its own work items: test_gpu_segments)
    try:
        _order_protocol(fused, args, cam, T, snaps, imgs)
    finally:
        fused.SEGMENTS = keep_seg


def _order_protocol(fused, args, cam, T, snaps, imgs):
    with torch.no_grad():
        for rep in range(fused.ORDER_REFRESH + 1):
            img, _, st = fused.forward(*args, cam)
            torch.cuda.synchronize()
            snaps.append(st.order.cpu().numpy().copy())
            imgs.append(img.cpu().numpy())
            assert rep == 0 or st.order is prev              # one buffer per camera
            assert st.order_by_work == (rep > 0)
            prev = st.order
    lens = (st.ranges[:, 1] - st.ranges[:, 0]).cpu().numpy()
    wk = lambda buf: buf[len(buf) - 2 * T:len(buf) - T]      # [order | work | walk]
    cont = st.contrib.cpu().numpy()
    gx = 640 // 16
    walked = np.array([cont[(t // gx) * 16:(t // gx) * 16 + 16, (t % gx) * 16:(t % gx) * 16 + 16].max() for t in range(T)])
    for rep, buf in enumerate(snaps):
        order, work = buf[:T], wk(buf)
        assert np.array_equal(np.sort(order), np.arange(T))
        assert (work >= 0).all() and (work[lens == 0] == 0).all() and work.max() <= 6 * lens.max()
        np.testing.assert_array_equal(work, wk(snaps[0]))    # same image, same work
        np.testing.assert_array_equal(buf[-T:], walked)      # the walk part: the tile's largest contributor index
        np.testing.assert_array_equal(imgs[rep], imgs[0])
    assert (np.diff(lens[snaps[0][:T]]) <= 0).all()          # first render: by list length
    by_work = snaps[1][:T]                                   # second render: by the first one's work, bins of 4
    assert (np.diff((wk(snaps[0]) // 4)[by_work]) <= 0).all()
    for rep in range(2, fused.ORDER_REFRESH + 1):
        if (rep + 1) % fused.ORDER_REFRESH == 0:             # refreshed (ties may land in another sequence)
            assert (np.diff((wk(snaps[0]) // 4)[snaps[rep][:T]]) <= 0).all()
        else:                                                # kept as it stands
            np.testing.assert_array_equal(snaps[rep][:T], snaps[rep - 1][:T])
    other = torch.cuda.Stream()                              # another stream: its own buffer
    with torch.cuda.stream(other), torch.no_grad():
        _, _, st2 = fused.forward(*args, cam)
    torch.cuda.synchronize()
    assert st2.order is not prev and not st2.order_by_work


def test_trainer_redoes_a_step_whose_view_outgrew_the_buffers():
    """``Trainer.step`` renders with deferred validation; when ``commit()`` reports an incomplete render (here:
    the learnt patch capacity is knocked down before the step) the step is redone and ends exactly where an
    undisturbed trainer ends."""
    from easygaussiansplatting_amd import fused, scene as S
    from easygaussiansplatting_amd.function import Camera, render
    from easygaussiansplatting_amd.trainer import Trainer
    sc = S.small_scene(4000, 160, 96, 48, seed=6)
    cams = [Camera.from_scene(c) for c in S.ring_cameras(sc.cam, 2, radius=5.0)]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    with torch.no_grad():
        gts = [render(dev(sc.pws), dev(sc.shs), dev(sc.alphas), dev(sc.scales), dev(sc.rots), c)[0] for c in cams]
    outs = []
    for disturb in (False, True):
        start = S.small_scene(4000, 160, 96, 48, seed=6)
        start.shs[:, :3] += 0.3
        tr = Trainer(start, cams, gts, max_steps=50, scene_size=4.0)
        losses = [tr.step([0, 1])]
        if disturb:
            cap = fused._ctx(torch.device("cuda", 0)).capacity
            for k in list(cap):
                if k[0] == 4000:
                    cap[k] = 100                       # far too small for the next renders
        losses += [tr.step([0, 1]) for _ in range(2)]
        assert tr.redone_steps == (1 if disturb else 0)
        outs.append((losses, {k: v.detach().cpu().numpy() for k, v in tr.params.items()}))
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=1e-5)
    for k in outs[0][1]:
        a, b = outs[0][1][k], outs[1][1][k]
        assert np.abs(a - b).max() < 1e-4 * max(1e-3, np.abs(a).max()), k


def test_backward_in_the_forward_dispatch_order(monkeypatch):
    """A camera rendered before is dispatched by its remembered work, and the backward pass then keeps that order
    (EGS_BWD_KEEP_FORWARD_ORDER) instead of sorting the tiles again: same gradients either way."""
    from easygaussiansplatting_amd import fused
    from easygaussiansplatting_amd.function import GSFunction
    args, cam = _scene(20000, 640, 368, 13)
    dl = torch.randn((3, 368, 640), device="cuda", generator=torch.Generator(device="cuda").manual_seed(23)) / (3 * 368 * 640)

    def grads():
        ps = [a.clone().requires_grad_(True) for a in args]
        ps[2] = ps[2].detach().reshape(-1, 1).clone().requires_grad_(True)
        us = torch.zeros((20000, 2), device="cuda", requires_grad=True)
        img, _ = GSFunction.apply(ps[0], ps[1], ps[2], ps[3], ps[4], us, cam)
        img.backward(dl)
        torch.cuda.synchronize()
        return [p.grad.cpu().numpy() for p in ps] + [us.grad.cpu().numpy()]
    grads()                                   # first render of this camera: leaves its per-tile work behind
    monkeypatch.setattr(fused, "REUSE_ORDER", False)
    a = grads()
    monkeypatch.setattr(fused, "REUSE_ORDER", True)
    b = grads()
    for x, y in zip(a, b):
        assert np.abs(x).max() > 0
        assert np.abs(x).max() > 0
        np.testing.assert_allclose(x, y, rtol=0, atol=2e-5 * np.abs(x).max())


def test_seven_op_splat_enqueues_ahead_too():
    """``gsplatcu.splat``: from the second call of a problem size on the draw stage is enqueued before P has been
    read (egs_splat_bin_mb / egs_splat_draw_dev); same five outputs as the synchronous sequence, also when the
    learnt capacity or the depth-key hint turn out too small, and the in-place cull of depths / areas is kept."""
    from easygaussiansplatting_amd import fused, scene as S
    from easygaussiansplatting_amd import gsplatcu as gsc
    from easygaussiansplatting_amd.function import Camera
    sc = S.small_scene(6000, 200, 120, 12, seed=8)
    cam = Camera.from_scene(sc.cam)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    pws, shs, alphas, scales, rots = dev(sc.pws), dev(sc.shs), dev(sc.alphas), dev(sc.scales), dev(sc.rots)
    us, pcs, depths0 = gsc.project(pws, cam.Rcw, cam.tcw, cam.fx, cam.fy, cam.cx, cam.cy, False)
    cov3ds = gsc.computeCov3D(rots, scales, depths0, False)[0]
    cov2ds = gsc.computeCov2D(cov3ds, pcs, cam.Rcw, depths0, cam.fx, cam.fy, 200, 120, False)[0]
    colors = gsc.sh2Color(shs, pws, cam.twc, False)[0]
    d1 = depths0.clone()
    cinv2ds, areas0 = gsc.inverseCov2D(cov2ds, d1, False)[:2]

    def run():
        d, a = d1.clone(), areas0.clone()
        out = gsc.splat(120, 200, us, cinv2ds, alphas, d, colors, a)
        torch.cuda.synchronize()
        return [x.cpu().numpy() for x in out] + [d.cpu().numpy(), a.cpu().numpy()]

    key = (6000, 200, 120)
    cap = fused._ctx(torch.device("cuda", 0)).capacity
    cap.pop(key, None)
    ref = run()                                              # no capacity yet: the synchronous sequence
    P = ref[4].shape[0]
    assert cap[key] > P > 1000
    for tweak in (None, "cap64", "cap-7", "hint"):
        if tweak == "cap64":
            cap[key] = 64
        elif tweak == "cap-7":
            cap[key] = P - 7
        elif tweak == "hint":
            gsc._set_key_bits(0, key, 1)
        got = run()
        for a, b in zip(ref, got):
            np.testing.assert_array_equal(a, b)
        assert cap[key] > P and 8 <= gsc._get_key_bits(0, key) <= 32
