"""What round 5 added, pinned against the ORACLE instead of against the previous kernel ("next round" 1):

(a) ``k_draw_bwd<SEG>`` -- one wave per segment of a long list, started from the segment-end states (G_s, T_end) the
    forward pass left -- on ``scene.skewed_scene(reset_alpha=True)`` at the production setting (L = 256 / split above
    1024): the four ``splatB``-level gradients (kernel.cu:809-950) of the Gaussians complete inside three windows of
    tiles -- one around the longest list -- against ``oracle.gs_oracle.draw_backward`` (float64 blend over the device's
    own float32 2D Gaussians), by the DEFAULT rule of tests/gradcheck.py (2e-4 of the maximum, median 1e-4: the
    reference's own ``check``, backward_cpu.py:61-65).  The UNSPLIT kernel goes through the very same check: with
    EGS_GRAD_STATS set both land in one file (profiles/r6_grad_errors.jsonl) -- which of the two is the accurate one is
    then a measurement, and ``tol_max=4e-4`` of tests/test_gpu_segments.py is not needed anywhere;
(b) the PUBLIC ``splat`` / ``splatB`` pair on the same scene: ``splatB`` is handed tensors only and REBUILDS the states
    (``egs_splat_bwd_seg``, rebuild) -- or, new in round 6, finds the forward's states kept for exactly these tensors;
(c) the tile lists of the opaque ``scene.skewed_scene()`` -- 160 screen-filling Gaussians of up to 7 820 tiles, the only
    place ``k_bin_emit``'s wave-per-rect emission runs at scale: ``patch_range_per_tile`` / ``gsid_per_patch`` of the
    seven-op surface BIT-EXACT against ``O.bin_tiles`` (createKeys, kernel.cu:46-80), ``check_culled_lists`` on the fused
    path, and a 320 x 1200 image (75 tile rows: the ``rb += 64`` second round of the emission) with screen-filling
    Gaussians, lists bit-exact and culled lists checked on every tile."""
import numpy as np
import pytest

from easygaussiansplatting_amd import scene as S
from oracle import gs_oracle as O
from tests.gradcheck import assert_grad_close, assert_grad_close_flips
from tests.test_gpu_parity import (_oracle_2d, check_culled_lists, complete_inside, dev, host, window_tiles)

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def fx():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from easygaussiansplatting_amd import fused, gsplatcu
    gsplatcu.set_policy("gsplatcu")
    keep = fused.SEGMENTS, fused.SEG_SPECULATE
    yield fused, gsplatcu
    fused.SEGMENTS, fused.SEG_SPECULATE = keep


def stages(gsc, sc):
    """the five per-Gaussian ops of the seven-op surface (forward_gpu.py:47-57) -> device tensors"""
    cam = sc.cam
    pws, rots, scales, alphas, shs = map(dev, (sc.pws, sc.rots, sc.scales, sc.alphas, sc.shs))
    Rcw, tcw, twc = dev(cam.Rcw), dev(cam.tcw), dev(cam.twc)
    us, pcs, depths = gsc.project(pws, Rcw, tcw, cam.fx, cam.fy, cam.cx, cam.cy, False)
    cov3 = gsc.computeCov3D(rots, scales, depths, False)[0]
    cov2 = gsc.computeCov2D(cov3, pcs, Rcw, depths, cam.fx, cam.fy, cam.width, cam.height, False)[0]
    col = gsc.sh2Color(shs, pws, twc, False)[0]
    cinv, areas = gsc.inverseCov2D(cov2, depths, False)
    return dict(us=us, cinv=cinv, alphas=alphas, depths=depths, col=col, areas=areas)


@pytest.fixture(scope="module")
def reset_scene(fx):
    """skewed_reset, its device stages and ONE oracle backward pass per (contrib, final_tau) it is asked for"""
    fused, gsc = fx
    sc = S.skewed_scene(reset_alpha=True)
    return sc, stages(gsc, sc)


def _windows(rg, gx, gy):
    """three 6 x 4-tile windows: around the longest list, in the image centre, on the ragged bottom tile row"""
    lens = rg[:, 1] - rg[:, 0]
    tl = int(np.argmax(lens))
    return np.unique(np.concatenate([window_tiles(gx, gy, tl % gx, tl // gx), window_tiles(gx, gy, gx // 2, gy // 2),
                                     window_tiles(gx, gy, gx // 5, gy - 1)]))


@pytest.mark.parametrize("how", ["unsplit", "handle", "public", "public_kept", "public_content"])
def test_skewed_reset_splatB_gradients_vs_oracle(fx, reset_scene, how):
    fused, gsc = fx
    from tests.oracle_parallel import draw_backward_tiles
    sc, g = reset_scene
    W, H = sc.cam.width, sc.cam.height
    dl = S.normal(3, 22, (3, H, W)).astype(np.float32) / (3 * H * W)
    fused.SEGMENTS = "0" if how == "unsplit" else "auto"
    # public: nothing kept (splatB rebuilds the states); public_kept: states matched by identity + version; public_content
    # (the default): states + a snapshot, what splatB is handed compared with it on the device
    keep_states = gsc.set_pair_states({"public_kept": True, "public_content": "content"}.get(how, False))
    try:
        out = None
        for _ in range(2):      # (the second call finds the walks of the first in the hint words: the steady state)
            d, a = g["depths"].clone(), g["areas"].clone()
            if how == "handle":
                out, h = gsc.splat_with_records(H, W, g["us"], g["cinv"], g["alphas"], d, g["col"], a)
            else:
                out, h = gsc.splat(H, W, g["us"], g["cinv"], g["alphas"], d, g["col"], a), None
        image, contrib, tau, ranges, gsid = out
        grads = gsc.splatB(H, W, g["us"], g["cinv"], g["alphas"], d, g["col"], contrib, tau, ranges, gsid, dev(dl),
                           records=h)
        if how == "handle":
            assert h is not None and h.seg is not None           # the forward split its long lists
        info = gsc.last_splatB_info()
        assert info["segments"] == (how != "unsplit"), info
        assert info["rebuilt"] == (how == "public"), info        # public: rebuilt; public_kept: the forward's states
        if how == "public_content":
            assert info["kept_states"] and not info["rebuilt"], info
            # the pair is a pure function of the VALUES it is handed: clones of all eight tensors find the states too ...
            cl = [t.clone() for t in (g["us"], g["cinv"], g["alphas"], g["col"], contrib, tau, ranges, gsid)]
            again = gsc.splatB(H, W, cl[0], cl[1], cl[2], d, cl[3], cl[4], cl[5], cl[6], cl[7], dev(dl))
            assert gsc.last_splatB_info()["kept_states"]
            for x, y in zip(grads, again):
                assert float((x - y).abs().max()) <= 2e-6 * float(x.abs().max())
            # ... and a write through .data (no version counter moves) is SEEN: everything again from the handed tensors,
            # equal to what a splatB that keeps nothing gives for them
            us2 = g["us"].clone()
            us2.data[::7] += 0.25
            got = gsc.splatB(H, W, us2, g["cinv"], g["alphas"], d, g["col"], contrib, tau, ranges, gsid, dev(dl))
            info2 = gsc.last_splatB_info()
            assert info2["rebuilt"] and not info2["kept_states"], info2
            gsc.set_pair_states(False)
            want = gsc.splatB(H, W, us2, g["cinv"], g["alphas"], d, g["col"], contrib, tau, ranges, gsid, dev(dl))
            gsc.set_pair_states("content")
            for x, y in zip(want, got):          # (the same kernels on the same data: the order of the float atomics)
                assert float((x - y).abs().max()) <= 1e-5 * float(x.abs().max())
        if how == "public_kept":
            assert info["kept_states"], info
            # the kept states survive a backward pass (it reads them): the same call again finds them and gives the same
            # gradients (float atomics in another order: 1e-6 of the maximum)
            again = gsc.splatB(H, W, g["us"], g["cinv"], g["alphas"], d, g["col"], contrib, tau, ranges, gsid, dev(dl))
            assert gsc.last_splatB_info()["kept_states"]
            for x, y in zip(grads, again):
                assert float((x - y).abs().max()) <= 2e-6 * float(x.abs().max())
            # alphas as [N,1] -- what GSFunction hands both calls (gsmodel.py:36, 67): splatB normalises it to [N] before it
            # forms its signature, and so must the entry (round 6: through GSFunction nothing ever matched)
            a1 = g["alphas"].reshape(-1, 1)
            d2, ar2 = g["depths"].clone(), g["areas"].clone()
            o2 = gsc.splat(H, W, g["us"], g["cinv"], a1, d2, g["col"], ar2)
            gsc.splatB(H, W, g["us"], g["cinv"], a1, d2, g["col"], o2[1], o2[2], o2[3], o2[4], dev(dl))
            assert gsc.last_splatB_info()["kept_states"]
            # an in-place write to one of the four tensors (a new version, whatever the values) and the states are
            # nobody's: rebuilt from ``contrib``, same gradients
            contrib.add_(0)
            rebuilt = gsc.splatB(H, W, g["us"], g["cinv"], g["alphas"], d, g["col"], contrib, tau, ranges, gsid, dev(dl))
            info2 = gsc.last_splatB_info()
            assert info2["rebuilt"] and not info2["kept_states"], info2
            for x, y, nm in zip(grads, rebuilt, ("dus", "dcinv", "dalpha", "dcolor")):
                assert_grad_close(host(y), host(x), "kept_vs_rebuilt:" + nm)
    finally:
        gsc.set_pair_states(keep_states)
    rg, gs = host(ranges), host(gsid)
    hcont, htau = host(contrib), host(tau)
    lens = rg[:, 1] - rg[:, 0]
    assert lens.max() > 10_000 and hcont.max() > 6_000           # lists to ~13 600, the longest walk > 8 000
    gx, gy = (W + 15) // 16, (H + 15) // 16
    sub = _windows(rg, gx, gy)
    assert int(np.argmax(lens)) in sub
    o = draw_backward_tiles(W, H, rg, gs, host(g["us"]), host(g["cinv"]), host(g["alphas"]), host(g["col"]), hcont, htau,
                            dl, tiles=sub)
    full = complete_inside(gs, rg, sub, sc.n)
    assert full.size > 1000, full.size
    for a, b, nm in zip(o[:4], grads, ("dus", "dcinv", "dalpha", "dcolor")):
        b = host(b).reshape(a.shape)
        r = assert_grad_close_flips(b[full], a[full], o[4][full], "skewed_reset_%s:%s" % (how, nm))   # DEFAULT rule
        assert r["n_big"] > 100, (nm, r)
    # A stated-precision MEASUREMENT next to the gate (EGS_GRAD_STATS only): the oracle above walks back from the DEVICE's
    # float32 final_tau by float64 divisions -- the reference's algorithm (kernel.cu:847-856), and the unsplit kernel's:
    # the two share that anchor and its per-pixel rounding (~sqrt(n) 6e-8 over an n-entry walk).  The segment kernel
    # starts every segment from the forward pass's own transmittance there: a different rounding history, which the
    # first moments of a Gaussian (dL/du = -cinv M1, a sum that cancels across its pixels) amplify.  The arbiter is the
    # oracle anchored on NOTHING of the device's: its own float64 forward pass over the same lists (a 3 x 2-tile window
    # around the longest list), then its backward pass from that.
    import os
    if os.environ.get("EGS_GRAD_STATS") and how in ("unsplit", "handle"):
        from tests.test_gpu_parity import record_grad_error
        tl = int(np.argmax(lens))
        win = window_tiles(gx, gy, tl % gx, tl // gx, 3, 2)
        hu, hc, ha, hcol = host(g["us"]), host(g["cinv"]), host(g["alphas"]), host(g["col"])
        _, c64, t64 = O.draw(W, H, rg, gs, hu, hc, ha, hcol, None, O.POLICY_G, tiles=win)
        o64 = draw_backward_tiles(W, H, rg, gs, hu, hc, ha, hcol, c64, t64, dl, tiles=win)
        fw = complete_inside(gs, rg, win, sc.n)
        for a, b, nm in zip(o64[:4], grads, ("dus", "dcinv", "dalpha", "dcolor")):
            record_grad_error("skewed_reset_%s_f64_forward_anchor:%s" % (how, nm), host(b).reshape(a.shape)[fw], a[fw],
                              o64[4][fw])


def _fillers(sc, count, smin, smax, seed=41):
    """every (n // count)-th Gaussian becomes a screen-filling one (as scene.skewed_scene's fillers)"""
    idx = (np.arange(count) * (sc.n // count) + sc.n // (2 * count)) % sc.n
    u = S.uniform01(seed, 1, (count, 3))
    sc.scales[idx] = (smin * (smax / smin) ** u).astype(np.float32)
    sc.pws[idx, 2] = (0.5 + 1.5 * S.uniform01(seed, 2, (count,))).astype(np.float32)
    sc.alphas[idx] = np.minimum(sc.alphas[idx], 0.2)
    return idx


def _lists_bit_exact(gsc, sc, g):
    W, H = sc.cam.width, sc.cam.height
    d, a = g["depths"].clone(), g["areas"].clone()
    image, contrib, tau, ranges, gsid = gsc.splat(H, W, g["us"], g["cinv"], g["alphas"], d, g["col"], a)
    o_d, o_a = host(g["depths"]).copy(), host(g["areas"]).copy()
    o_rg, o_gs, o_rects, o_counts = O.bin_tiles(host(g["us"]), o_a, o_d, W, H, O.POLICY_G)
    assert np.array_equal(host(ranges), o_rg), "patch_range_per_tile differs from the oracle's"
    assert np.array_equal(host(gsid), o_gs), "gsid_per_patch differs from the oracle's"
    assert np.array_equal(host(d), o_d) and np.array_equal(host(a), o_a)       # the in-place contract (kernel.cu:114-119)
    return o_rects, o_counts, host(ranges), o_gs


def test_skewed_opaque_scene_lists_bit_exact_and_culled_lists(fx):
    """(c) the opaque skewed scene at 1080p: the reference's lists bit for bit on the seven-op surface (8.9 M patches,
    rects of up to 120 x 68 tiles), and the fused path's footprint-culled lists on the tiles of the three densest
    windows + 16 tiles every filler covers."""
    fused, gsc = fx
    from easygaussiansplatting_amd.function import Camera
    sc = S.skewed_scene()
    W, H = sc.cam.width, sc.cam.height
    g = stages(gsc, sc)
    o_rects, o_counts, rg, _ = _lists_bit_exact(gsc, sc, g)
    wh = o_rects[:, 2:4].astype(np.int64) - o_rects[:, 0:2].astype(np.int64)
    assert ((wh[:, 1] > 64) & (o_counts > 0)).sum() >= 20            # rects taller than 64 tile rows exist ...
    assert o_counts.max() > 7000                                      # ... and screen-filling ones
    del g
    fused.SEGMENTS = "auto"
    with torch.no_grad():
        _, _, st = fused.forward(dev(sc.pws), dev(sc.shs), dev(sc.alphas).reshape(-1, 1), dev(sc.scales), dev(sc.rots),
                                 Camera.from_scene(sc.cam), need_grad=False)
    assert st.culled
    gx, gy = (W + 15) // 16, (H + 15) // 16
    lens = rg[:, 1] - rg[:, 0]
    tl = int(np.argmax(lens))
    tiles = np.unique(np.concatenate([window_tiles(gx, gy, tl % gx, tl // gx, 3, 2), [0, gx - 1, (gy - 1) * gx, gx * gy - 1],
                                      (S.uniform01(4, 3, (12,)) * gx * gy).astype(np.int64)]))
    o_us, o_ci, _, o_depths, o_areas = _oracle_2d(sc, sc.cam)
    d_marked = o_depths.astype(np.float32).copy()
    rects64, _ = O.get_rects(o_us.astype(np.float32), o_areas.copy(), d_marked, W, H, O.POLICY_G)
    dropped, kept, bdev, btrue = check_culled_lists(st, tiles, o_us, o_ci, sc.alphas.astype(np.float64), host(st.depths),
                                                    rects64.astype(np.int64), W)
    assert kept > 5000 and dropped > 0 and btrue <= bdev <= 1.35 * btrue, (dropped, kept, bdev, btrue)


@pytest.mark.parametrize("W,H", [(320, 1200), (2080, 96)])
def test_wave_per_rect_emission_beyond_64_tile_rows(fx, W, H):
    """(c) ``k_bin_emit`` emits a footprint-culled rect beyond 8 x 8 tiles with its WAVE, 64 tile rows per round
    (csrc/egs_bin.hip): a 320 x 1200 image has 75 tile rows, so a screen-filling Gaussian needs the second round; the
    2080 x 96 one has 130-tile-wide rows (the inner ``tx += 64`` loop).  Seven-op lists bit-exact against the oracle,
    the fused path's culled lists checked on EVERY tile, image against the oracle's pipeline."""
    fused, gsc = fx
    from easygaussiansplatting_amd.function import Camera
    sc = S.small_scene(6000, W, H, 12, seed=123)
    sc.cam = type(sc.cam)(W, H, 300.0, 300.0, W / 2.0, H / 2.0, sc.cam.Rcw, sc.cam.tcw)
    idx = _fillers(sc, 24, 4.0, 16.0)
    sc.scales[idx[::3], 0] *= 0.05                       # a third of them thin: tall / wide rects with narrow footprints
    g = stages(gsc, sc)
    o_rects, o_counts, rg, gs = _lists_bit_exact(gsc, sc, g)
    wh = o_rects[:, 2:4].astype(np.int64) - o_rects[:, 0:2].astype(np.int64)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    assert (wh[idx, 1] == gy).sum() >= 6 and (wh[idx, 0] == gx).sum() >= 6, (wh[idx].max(0), gx, gy)
    fused.SEGMENTS = "auto"
    with torch.no_grad():
        img, _, st = fused.forward(dev(sc.pws), dev(sc.shs), dev(sc.alphas).reshape(-1, 1), dev(sc.scales), dev(sc.rots),
                                   Camera.from_scene(sc.cam), need_grad=False)
    assert st.culled
    o_us, o_ci, o_col, o_depths, o_areas = _oracle_2d(sc, sc.cam)
    d_marked = o_depths.astype(np.float32).copy()
    rects64, _ = O.get_rects(o_us.astype(np.float32), o_areas.copy(), d_marked, W, H, O.POLICY_G)
    rf = host(st.ranges)
    tiles = np.nonzero(rf[:, 1] > rf[:, 0])[0]
    dropped, kept, bdev, btrue = check_culled_lists(st, tiles, o_us, o_ci, sc.alphas.astype(np.float64), host(st.depths),
                                                    rects64.astype(np.int64), W)
    assert dropped > 0 and kept > 10000 and btrue <= bdev <= 1.6 * btrue, (dropped, kept, bdev, btrue)
    # the image: the oracle's blend over the REFERENCE's (unculled) lists from its own float64 2D Gaussians
    o_img, o_cont, _ = O.draw(W, H, rg, gs, o_us, o_ci, sc.alphas.astype(np.float64), o_col, None, O.POLICY_G)
    d = np.abs(host(img) - o_img).max(0)
    assert (d >= 1e-4).sum() <= 8 and d.max() < 5e-3, ((d >= 1e-4).sum(), d.max())      # threshold flips, counted
