"""BASELINE configs[3]'s workload on ONE GPU: the 8 ring cameras of the 1 M-Gaussian scene at 1920x1080, rendered
one after another through the HIP path (``GSFunction`` in mode "fused", the path bench.py times) and compared with
the oracle view by view -- the counterpart of the reference's one-view-per-step loop (train.py:48-57) for the views
a multi-GPU run hands to ranks 1..7.  Views 1-7 differ from view 0 in everything the binning depends on: depth range
(1.8 .. 10.2 m against 4 .. 8 m: one more significant depth-key bit), cull pattern (up to 11 % of the Gaussians leave
the image), patch count and list lengths.

Per view:   * invariants of the tile lists (they tile the patch array, every list is sorted by (depth key, index),
              every Gaussian appears exactly rect-many times);
            * the image on sampled tiles against ``O.draw`` fed with the ORACLE's own float64 2D Gaussians, and ALL
              8160 tiles against the all-tile digest of the pinned oracle (fixture G11: mean RGB, mean tau, finished
              pixels per tile; 64 full tiles for views 0 and 3);
            * the five parameter gradients (+ dL/du) of the Gaussians complete inside three contiguous windows of
              tiles (3 000+ per view) against ``O.draw_backward`` + ``O.chain_rule`` under the RELATIVE rule of
              tests/gradcheck.py (threshold-flip Gaussians named by the oracle and counted).
Then:       * gradients accumulated by autograd over the 8 views in one process == sum of the per-view gradients;
            * the 8 views dealt to three HIP streams (``ViewStreams``, deferred validation) give the same sum;
            * the same 8 views through the overlapped exchange path (``ChunkedExchange`` on a one-rank process group,
              what ``EGS_FORCE_EXCHANGE=1 python bench.py --overlap-exchange`` runs; chunk counts 2, 4, 8 in turn) give the same mean.
"""
import contextlib
import os
import socket

import numpy as np
import pytest

from easygaussiansplatting_amd import scene as S
from oracle import gs_oracle as O
from tests.conftest import load_golden
from tests.gradcheck import assert_grad_close_flips
from tests.test_gpu_parity import (LIKE_MARGIN, LIKE_NEAR_FRAC, LIKE_U_ULPS, _oracle_2d, check_against_g11, check_culled_lists, complete_inside,
                                   dev, gradient_windows, host)

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

N_VIEWS = 8
NAMES = ("pws", "shs", "alphas", "scales", "rots")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _render_grads(P, cam, dl, GSFunction, exchange=None):
    """One forward + backward on fresh leaves -> (image, mask, {name: grad}, dus)."""
    leaves = {k: v.detach().requires_grad_(True) for k, v in P.items()}
    us0 = torch.zeros((P["pws"].shape[0], 2), device="cuda", requires_grad=True)
    if exchange is None:
        image, mask = GSFunction.apply(*[leaves[k] for k in NAMES], us0, cam)
        image.backward(dl)
    else:
        with exchange.attach():
            image, mask = GSFunction.apply(*[leaves[k] for k in NAMES], us0, cam)
            image.backward(dl)
        assert exchange.finish([leaves[k] for k in NAMES])      # verifies that .grad IS the exchanged storage
    return image.detach(), mask, {k: leaves[k].grad for k in NAMES}, us0.grad


def test_eight_ring_views_full_size():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.distributed as dist
    from easygaussiansplatting_amd import dist_views as DV
    from easygaussiansplatting_amd import fused, gsplatcu
    from easygaussiansplatting_amd.function import Camera, GSFunction
    gsplatcu.set_policy("gsplatcu")
    GSFunction.mode = "fused"
    sc = S.big_scene()
    W, H = sc.cam.width, sc.cam.height
    cams_np = S.ring_cameras(sc.cam, N_VIEWS)
    cams = [Camera.from_scene(c) for c in cams_np]
    P = dict(pws=dev(sc.pws), shs=dev(sc.shs), alphas=dev(sc.alphas).reshape(-1, 1), scales=dev(sc.scales),
             rots=dev(sc.rots))
    alphas64 = sc.alphas.astype(np.float64)
    gx = (W + 15) // 16
    dls = [dev(S.normal(8, 10 + v, (3, H, W)).astype(np.float32) / (H * W)) for v in range(N_VIEWS)]
    g11 = load_golden("g11_policy_g_1m_digest.npz")
    per_view = []          # per-view gradients (device tensors), for the accumulation checks below
    stats = []
    for v in range(N_VIEWS):
        cam, cnp = cams[v], cams_np[v]
        with torch.no_grad():
            # (need_grad=True: the kernel instance GSFunction.forward runs -- bit-identical renders)
            img_t, mask_t, st = fused.forward(P["pws"], P["shs"], P["alphas"], P["scales"], P["rots"], cam,
                                              need_grad=True)
        npatch = st.patch_count()
        rg, gs = host(st.ranges), host(st.gaussian_ids())
        hcont, htau, hdepth = host(st.contrib), host(st.final_tau), host(st.depths)
        # ---- invariants of the tile lists (size-independent properties)
        lens = rg[:, 1] - rg[:, 0]
        T = rg.shape[0]
        assert lens.min() >= 0 and int(lens.sum()) == npatch
        nz = lens > 0
        assert np.array_equal(rg[nz, 0][1:], rg[nz, 1][:-1]) and rg[nz, 0][0] == 0
        keys = O.depth_keys(hdepth, O.POLICY_G).astype(np.int64)
        comp = keys[gs] * (1 << 21) + gs
        inner = np.ones(npatch, bool); inner[rg[nz, 0]] = False
        assert (np.diff(comp)[inner[1:]] > 0).all(), "a tile list is not sorted by (depth key, index) in view %d" % v
        o_us, o_ci, o_col, o_depths, o_areas = _oracle_2d(sc, cnp)
        d_marked = o_depths.astype(np.float32).copy()
        o_rects, counts = O.get_rects(o_us.astype(np.float32), o_areas.copy(), d_marked, W, H, O.POLICY_G)
        allp = np.bincount(gs, minlength=sc.n)
        # The lists are footprint-culled: a Gaussian appears for at most the tiles of its rect.  (float32 (device) vs
        # float64 (oracle) centres / radii may disagree on a rect whose edge sits on a tile border: a handful in a
        # million may exceed the oracle's count.)
        assert st.culled and (allp > counts).sum() <= 300, (v, (allp > counts).sum())
        assert 0.80 * counts.sum() < npatch < 0.95 * counts.sum(), (v, npatch, counts.sum())
        hmask = host(mask_t)
        assert (hmask != (d_marked > 0.2)).sum() <= 16
        stats.append((npatch, int(lens.max()), int((~hmask).sum()), int(keys.max()).bit_length()))
        # ---- the gradient render of this view (the path bench.py times)
        image, mask, g, dus = _render_grads(P, cam, dls[v], GSFunction)
        assert torch.equal(image, img_t) and torch.equal(mask, mask_t)
        per_view.append({k: g[k].clone() for k in NAMES})
        him = host(image)
        # ALL tiles of this view against the all-tile digest of the pinned oracle (fixture G11)
        check_against_g11(g11, v, him, htau, W, H, label="ring_view%d" % v, culled_lens=lens)
        sel = (S.uniform01(40 + v, 2, (8,)) * T).astype(np.int64)
        sel = np.array([t for t in sel if lens[t] > 0] or [int(np.argmax(lens))])
        dropped, kept, bdev, btrue = check_culled_lists(st, sel[:4], o_us, o_ci, alphas64, hdepth,
                                                        o_rects.astype(np.int64), W)
        assert dropped > 0 and btrue <= bdev, (v, dropped, kept, bdev, btrue)
        o_img, o_cont, o_tau = O.draw(W, H, rg, gs, o_us, o_ci, alphas64, o_col, None, O.POLICY_G, tiles=sel)
        nflip = 0
        for t in sel:
            ty, tx = divmod(int(t), gx)
            ys = slice(ty * 16, min(ty * 16 + 16, H)); xs = slice(tx * 16, tx * 16 + 16)
            d = np.abs(him[:, ys, xs] - o_img[:, ys, xs]).max(0)
            flip = (hcont[ys, xs] != o_cont[ys, xs]) | (d >= 1e-4)
            nflip += int(flip.sum())
            assert d[~flip].max() < 1e-4 and d.max() < 5e-3, (v, int(t), d.max())
        assert nflip <= 8, (v, nflip)                     # alpha' >= 0.002 / tau < 1e-4 threshold flips, fp32 vs fp64
        # ---- gradients of the Gaussians complete inside two contiguous windows of tiles (image centre and around the
        # longest list): every one of them gets its whole gradient there -- a thousand or more per view, with a
        # reference gradient that is not zero (view 2 looks along the scene's long axis: most of a random tile's
        # Gaussians are hidden behind saturated pixels and receive none)
        sub = gradient_windows(rg, gx, (H + 15) // 16, count=3)[:]
        full = complete_inside(gs, rg, sub, sc.n)
        assert full.size > 500, (v, full.size)
        dl64 = host(dls[v]).astype(np.float64)
        near = np.zeros(sc.n, bool)
        # like for like (tests/test_gpu_parity.py::test_full_size_fused_and_raw_paths): the oracle's stages in the
        # device's float32, its blend and chain rule in float64 -- the default rule of tests/gradcheck.py
        q_us, q_ci, q_col, _, _ = _oracle_2d(sc, cnp, dtype=np.float32)
        o_g2 = O.draw_backward(W, H, rg, gs, q_us, q_ci, alphas64, q_col, hcont, htau, dl64, None, O.POLICY_G,
                               tiles=sub, near_out=near, near_margin=LIKE_MARGIN, near_u_ulps=LIKE_U_ULPS)
        _, _, _, _, J = _oracle_2d(sc, cnp, full, True, np.float32)
        og = O.chain_rule(o_g2[0][full], o_g2[1][full], o_g2[2][full], o_g2[3][full], cnp.Rcw, J)
        want = dict(pws=og["dpws"], shs=og["dshs"], alphas=og["dalphas"][:, None], scales=og["dscales"],
                    rots=og["drots"], us=o_g2[0][full])
        got = {k: host(g[k])[full] for k in NAMES} | {"us": host(dus)[full]}
        for k in want:
            assert got[k].shape == want[k].shape, (v, k)
            r = assert_grad_close_flips(got[k], want[k], near[full], "ring_view%d_f32_stages:%s" % (v, k),
                                        near_frac=LIKE_NEAR_FRAC)
            assert r["n_big"] > 50, (v, k, r)
    # the views really are different workloads
    assert len({s[0] for s in stats}) == N_VIEWS, stats
    assert max(s[3] for s in stats) > stats[0][3], stats      # a ring view needs more depth-key bits than view 0

    # ---- 8 views accumulated by autograd in one process (what a rank with 8 local views holds before the exchange)
    leaves = {k: v.detach().requires_grad_(True) for k, v in P.items()}
    us0 = torch.zeros((sc.n, 2), device="cuda", requires_grad=True)
    for v in range(N_VIEWS):
        image, _ = GSFunction.apply(*[leaves[k] for k in NAMES], us0, cams[v])
        image.backward(dls[v])
    flat = fused.flat_grad_buffer([leaves[k] for k in NAMES])
    assert flat is not None and flat.numel() >= 59 * sc.n        # still ONE buffer: one collective for 8 views
    for k in NAMES:
        tot = sum(pv[k] for pv in per_view)
        scale = float(tot.abs().max())
        assert float((leaves[k].grad - tot).abs().max()) <= 1e-5 * scale, k      # atomics order only

    # ---- the same 8 views dealt to three streams, deferred validation (what ``bench.py --views-per-rank 8`` and the
    # ``ring_views_8`` leg of the default run time): the sum over the lanes' accumulators
    vleaves = [P[k].detach().requires_grad_(True) for k in NAMES]
    vs = DV.ViewStreams(vleaves, 3)
    uss = [torch.zeros((sc.n, 2), device="cuda", requires_grad=True) for _ in range(3)]
    # third step: the SH gradient kept factored (``FactoredShGrad``: every view leaves dL/dcolour [N,3], one kernel forms
    # the step's rows -- what the ``ring_views_8`` leg and ``--views-per-rank 8`` run by default)
    fx = DV.FactoredShGrad(N_VIEWS)
    for rep in range(3):                       # (the second step starts from empty accumulators, capacities learnt)
        for t in vleaves:
            t.grad = None
        with fused.deferred() as d:
            vs.begin()
            with fused.accumulate_in_kernel(), (fx.attach() if rep == 2 else contextlib.nullcontext()):
                for v in range(N_VIEWS):
                    with vs.lane(v) as lv:
                        image, _ = GSFunction.apply(*lv, uss[vs.lane_index(v)], cams[v])
                        image.backward(dls[v])
            vs.finish()
            if rep == 2:
                assert vleaves[1].grad is None
                fx.finish(vleaves[0], vleaves[1])
            assert not d.commit()
        torch.cuda.synchronize()
        assert fused.flat_grad_buffer(vleaves if rep < 2 else vleaves[:1] + vleaves[2:]) is not None
        for k, t in zip(NAMES, vleaves):
            ref = leaves[k].grad
            assert float((t.grad - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), (rep, k)

    # ---- the same views through the overlapped exchange (one-rank group: the collectives run, sums are unchanged)
    started = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ["MASTER_PORT"] = str(_free_port())
        try:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        except Exception:
            dist.init_process_group("gloo", rank=0, world_size=1)
        started = True
    try:
        mean = {k: torch.zeros_like(P[k]) for k in NAMES}
        for v in range(N_VIEWS):
            ex = DV.ChunkedExchange(world=1, chunks=(2, 4, 8)[v % 3])
            _, _, g, _ = _render_grads(P, cams[v], dls[v], GSFunction, exchange=ex)
            assert ex.used
            for k in NAMES:
                mean[k] += g[k] / N_VIEWS
        torch.cuda.synchronize()
        for k in NAMES:
            ref = leaves[k].grad / N_VIEWS
            assert float((mean[k] - ref).abs().max()) <= 1e-5 * float(ref.abs().max()), k
        # a second backward inside one attach() must be refused (its .grad += would race the in-flight all-reduce)
        ex = DV.ChunkedExchange(world=1)
        lv = {k: v.detach().requires_grad_(True) for k, v in P.items()}
        u = torch.zeros((sc.n, 2), device="cuda", requires_grad=True)
        with ex.attach():
            img, _ = GSFunction.apply(*[lv[k] for k in NAMES], u, cams[0])
            img.backward(dls[0])
            img2, _ = GSFunction.apply(*[lv[k] for k in NAMES], u, cams[1])
            with pytest.raises(RuntimeError, match="second backward"):
                img2.backward(dls[1])
        ex.finish()
        fused.commit()
    finally:
        if started:
            dist.destroy_process_group()
