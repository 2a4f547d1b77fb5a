"""world_size-2 gloo test of the one-view-per-rank gradient exchange
(easygaussiansplatting_amd/dist_views.py).  Per-view gradients
come from the CPU oracle; the property pinned is the one the 8-GPU path relies
on: N-view gradient accumulation on one process == the all-reduced mean of one
view per rank."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist          # noqa: E402
import torch.multiprocessing as mp        # noqa: E402

from easygaussiansplatting_amd import dist_views as DV   # noqa: E402
from easygaussiansplatting_amd import scene as S         # noqa: E402
from oracle import gs_oracle as O                        # noqa: E402


def view_grads(sc, cam, seed):
    """Oracle forward+backward of one view -> the 5 parameter gradients + density stats."""
    P = O.POLICY_G
    us, pcs, depths, du = O.project(sc.pws, cam.Rcw, cam.tcw, cam.fx, cam.fy, cam.cx, cam.cy, P, True)
    c3, dq, ds = O.compute_cov3d(sc.rots, sc.scales, depths, P, True)
    c2, d3, dpc = O.compute_cov2d(c3, pcs, cam.Rcw, depths, cam.fx, cam.fy, cam.width, cam.height, P, True)
    col, dsh, dpw = O.sh2color(sc.shs, sc.pws, cam.twc, True)
    ci, areas, dci = O.inverse_cov2d(c2, depths, P, True)
    img, cont, tau, ranges, gsid = O.splat(cam.height, cam.width, us, ci, sc.alphas, depths, col, areas, P)
    dl = S.normal(seed, 3, img.shape) / img.size
    dus, dcinv, dal, dcol = O.draw_backward(cam.width, cam.height, ranges, gsid, us, ci, sc.alphas, col, cont, tau,
                                            dl, None, P)
    J = dict(dcinv2d_dcov2ds=dci, dcov2d_dcov3ds=d3, dcov3d_drots=dq, dcov3d_dscales=ds, dcolor_dshs=dsh,
             du_dpcs=du, dcov2d_dpcs=dpc, dcolor_dpws=dpw)
    g = O.chain_rule(dus, dcinv, dal, dcol, cam.Rcw, J)
    grads = dict(pws=g["dpws"], shs=g["dshs"], alphas=g["dalphas"][:, None], scales=g["dscales"], rots=g["drots"])
    grads["_dcolor"] = dcol        # (not a parameter gradient: the factor FactoredShGrad exchanges)
    return grads, dus, depths > 0.2


def _scene():
    sc = S.small_scene(300, 64, 48, 12, seed=11)
    cams = S.ring_cameras(sc.cam, 2, radius=5.0)
    return sc, cams


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc, cams = _scene()
        assert DV.views_for_rank(2, rank, world) == [rank]
        grads, dus, mask = view_grads(sc, cams[rank], seed=100)
        params = {}
        for k in DV.PARAM_ORDER:
            p = torch.zeros(grads[k].shape, dtype=torch.float64, requires_grad=True)
            p.grad = torch.from_numpy(np.ascontiguousarray(grads[k]))
            params[k] = p
        DV.exchange_gradients(params)
        gn, cnt = DV.density_stats(torch.from_numpy(dus), torch.from_numpy(mask))
        q.put((rank, {k: params[k].grad.numpy().copy() for k in DV.PARAM_ORDER}, gn.numpy().copy(),
               cnt.numpy().copy()))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_two_rank_exchange_equals_single_process_accumulation():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sc, cams = _scene()
    per_view = [view_grads(sc, c, seed=100) for c in cams]
    for k in DV.PARAM_ORDER:
        mean = (per_view[0][0][k] + per_view[1][0][k]) / 2
        for _, g, _, _ in res:
            np.testing.assert_allclose(g[k], mean, rtol=1e-12, atol=1e-15)
        assert np.abs(mean).max() > 0
    norm_sum = sum(np.where(m, np.linalg.norm(d, axis=1), 0) for _, d, m in per_view)
    cnt_sum = sum(m.astype(np.int32) for _, _, m in per_view)
    for _, _, gn, cnt in res:
        np.testing.assert_allclose(gn, norm_sum, rtol=1e-12)
        assert np.array_equal(cnt, cnt_sum)


def test_single_process_is_a_noop():
    t = torch.ones(4)
    DV.allreduce_mean_([t])
    assert torch.equal(t, torch.ones(4))
    assert DV.grad_exchange_bytes(1_000_000) == 236_000_000
    assert DV.views_for_rank(8, 3, 8) == [3] and DV.views_for_rank(8, 1, 2) == [1, 3, 5, 7]


def _worker_flat(rank, world, port, q):
    """Gradients laid out the way the fused backward allocates them: 16-B aligned slices of one buffer."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, widths = 37, (3, 48, 1, 3, 4)
        starts, at = [], 0
        for w in widths:
            starts.append(at)
            at += (n * w + 3) // 4 * 4
        flat = torch.full((at,), float("nan"), dtype=torch.float32)      # padding stays garbage
        params = {}
        for k, a, w in zip(DV.PARAM_ORDER, starts, widths):
            p = torch.zeros(n, w, requires_grad=True)
            g = flat[a:a + n * w].view(n, w)
            g.copy_(torch.arange(n * w, dtype=torch.float32).view(n, w) * (rank + 1) + a)
            p.grad = g
            params[k] = p
        calls = []
        real = dist.all_reduce
        dist.all_reduce = lambda t, *a, **k: (calls.append(t.numel()), real(t, *a, **k))[1]
        try:
            DV.exchange_gradients(params)
        finally:
            dist.all_reduce = real
        q.put((rank, calls, {k: params[k].grad.numpy().copy() for k in DV.PARAM_ORDER}, starts))
    finally:
        dist.destroy_process_group()


def test_two_rank_exchange_of_a_flat_gradient_buffer_is_one_collective():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_flat, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n, widths = 37, (3, 48, 1, 3, 4)
    for rank, calls, grads, starts in res:
        assert len(calls) == 1 and calls[0] >= 59 * n          # ONE all-reduce over the whole buffer
        for k, a, w in zip(DV.PARAM_ORDER, starts, widths):
            want = np.arange(n * w, dtype=np.float32).reshape(n, w) * 1.5 + a      # mean of x1 and x2
            np.testing.assert_allclose(grads[k], want, rtol=1e-6)


def _worker_factored(rank, world, port, q):
    """The SH gradient exchanged in its factored form: every rank puts dL/dcolour [N,3] and the camera centre of ITS
    view into a ``FactoredShGrad`` row; ``gathered()`` is the collective of ``finish()`` (the kernel that forms the rows
    from it is HIP: tests/test_gpu_factored_sh.py, tests/test_gpu_dist_views.py)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc, cams = _scene()
        grads, _, _ = view_grads(sc, cams[rank], seed=100)
        n = sc.pws.shape[0]
        fx = DV.FactoredShGrad(views=2)                 # two rows per rank, this rank fills one: the other counts as 0

        class Cam:
            twc = torch.from_numpy(np.asarray(cams[rank].twc, np.float32).reshape(3))
        slot = fx.slot(n, sc.shs.shape[1], Cam)
        assert slot.shape == (n, 3) and fx.rows.shape == (2, DV.FactoredShGrad.row_stride(n))
        slot.copy_(torch.from_numpy(grads["_dcolor"].astype(np.float32)))
        fx.rows[0, 3 * n:3 * n + 3] = Cam.twc          # (on the GPU the backward kernel stores it: EGS_BWD_FACTORED_SH)
        rows, w = fx.gathered()
        q.put((rank, rows.numpy().copy(), w))
    finally:
        dist.destroy_process_group()


def test_two_rank_factored_sh_exchange_carries_the_rows():
    """eq (5) of backward.md (gsmodel.py:84-85) is an outer product: from the all-gathered (dL/dcolour, camera centre)
    of both views every rank can form the mean of the two views' dL/dshs rows -- 12 B per Gaussian and view on the wire
    instead of 4 sh_dim per Gaussian."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_factored, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sc, cams = _scene()
    n, K = sc.shs.shape
    per_view = [view_grads(sc, c, seed=100)[0] for c in cams]
    want = (per_view[0]["shs"] + per_view[1]["shs"]) / 2
    assert np.abs(want).max() > 0
    for rank, rows, w in res:
        assert w == 2 and rows.shape == (4, DV.FactoredShGrad.row_stride(n))
        np.testing.assert_array_equal(rows, res[0][1])                  # every rank holds the same rows
        assert not rows[1, :3 * n].any() and not rows[3, :3 * n].any()  # the rows nobody filled
        got = np.zeros((n, K))
        for v in range(rows.shape[0]):
            g = rows[v, :3 * n].reshape(n, 3).astype(np.float64)
            d = sc.pws - rows[v, 3 * n:3 * n + 3].astype(np.float64)
            B = O.sh_basis(d / np.sqrt((d * d).sum(1))[:, None], K // 3)[:, :K // 3]
            got += (B[:, :, None] * g[:, None, :]).reshape(n, K)        # sh[i, 3 c + rgb]
        np.testing.assert_allclose(got / w, want, rtol=0, atol=2e-6 * np.abs(want).max())   # (fp32 on the wire)
    assert DV.factored_exchange_pays(8, 1) and DV.factored_exchange_pays(2, 1) and not DV.factored_exchange_pays(8, 4)
    assert DV.factored_exchange_pays(1, 8) and not DV.factored_exchange_pays(1, 1)
    assert DV.factored_exchange_pays(4, 1, 12) and not DV.factored_exchange_pays(8, 1, 3)
