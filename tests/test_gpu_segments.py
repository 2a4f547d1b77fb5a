"""Long tile lists split over several waves (include/egs_hip.h ``egs_splat_draw_rec_seg``; the reference spends 256
threads on a tile, kernel.cu:152-271 -- here a tile is one wave64 and a long list is walked in segments).

What is checked: the segment path -- every kind of work item (DIRECT tiles, SPEC segments blended from tau = 1 with the
re-walk of the pixels that finish inside them, COMPOSE with sequential continuation) and the backward pass that walks
every segment with a wave of its own -- gives the images, contributor counts, final transmittances and parameter
gradients of the unsplit kernels (which the rest of the suite pins against the oracle), and the oracle's on sampled
tiles including the longest list.  Small segments (64 / 128 entries) put dozens of segment boundaries into ordinary
scenes; the heavy-tailed 1080p scene runs at the production setting."""
import ctypes as C

import numpy as np
import pytest

from easygaussiansplatting_amd import scene as S
from oracle import gs_oracle as O
from tests.gradcheck import assert_grad_close

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def fx():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from easygaussiansplatting_amd import _lib, fused, gsplatcu
    gsplatcu.set_policy("gsplatcu")
    lib = _lib.load()
    before = (C.c_int * 2)()
    _lib.check(lib.egs_seg_config(0, 0, before))
    keep = fused.SEGMENTS
    yield fused, lib
    fused.SEGMENTS = keep
    _lib.check(lib.egs_seg_config(before[0], before[1], None))


def dev(a, dtype=np.float32):
    return torch.from_numpy(np.ascontiguousarray(a, dtype)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def run(fused, sc, cam, dl, renders=1):
    """forward + backward of ``sc`` through GSFunction (fused); ``renders`` > 1: the same camera object again, so that
    the later renders find the walk lengths of the earlier ones (SPEC items).  -> dict of the last render."""
    from easygaussiansplatting_amd.function import GSFunction
    GSFunction.mode = "fused"
    out = None
    for _ in range(renders):
        P = dict(pws=dev(sc.pws), shs=dev(sc.shs), alphas=dev(sc.alphas).reshape(-1, 1), scales=dev(sc.scales),
                 rots=dev(sc.rots))
        for p in P.values():
            p.requires_grad_(True)
        us0 = torch.zeros((sc.n, 2), device="cuda", requires_grad=True)
        image, mask = GSFunction.apply(P["pws"], P["shs"], P["alphas"], P["scales"], P["rots"], us0, cam)
        image.backward(dl)
        out = dict(image=host(image), grads={k: host(v.grad) for k, v in P.items()} | {"us": host(us0.grad)})
    # the state of an identical render (same kernels): contrib / final_tau / lists
    with torch.no_grad():
        d = {k: v.detach() for k, v in P.items()}
        img2, _, st = fused.forward(d["pws"], d["shs"], d["alphas"], d["scales"], d["rots"], cam, need_grad=True)
    out.update(contrib=host(st.contrib), tau=host(st.final_tau), ranges=host(st.ranges), ids=host(st.gaussian_ids()),
               image2=host(img2), seg=st.seg is not None)
    return out


def compare(a, b, label, flips=8, tol_max=2e-4):
    """segment path ``a`` against the unsplit kernels ``b``: same lists; the image differs by the rounding of the
    composed sum; a pixel may stop one entry earlier or later where T * tau_local straddles tau_stop (counted)."""
    assert np.array_equal(a["ranges"], b["ranges"]) and np.array_equal(a["ids"], b["ids"])
    d = np.abs(a["image"] - b["image"]).max(0)
    flip = (a["contrib"] != b["contrib"])
    assert flip.sum() <= flips, (label, int(flip.sum()))
    assert d[~flip].max() < 2e-5 and d.max() < 2e-3, (label, d[~flip].max(), d.max())
    assert np.abs(a["tau"] - b["tau"])[~flip].max() < 1e-5, label
    # (a later render of the same camera: more of its segments are SPEC items -- blended from tau = 1 and scaled by T_s --
    # where the earlier one walked on from T_s: the rounding of T_s C_s against a running sum)
    assert np.abs(a["image2"] - a["image"]).max() < 1e-5, label
    for k in a["grads"]:
        # (the unsplit backward pass un-does tau by thousands of divisions from final_tau; a segment starts from the
        # forward pass's own transmittance at its end: the two differ by that accumulated rounding, 6e-5 of the maximum
        # on 3 000-entry lists of opacity 0.01)
        assert_grad_close(a["grads"][k], b["grads"][k], label + ":" + k, tol_max=tol_max, med_rel=2e-5, max_rel=2e-3,
                          outliers=max(2, flips))


@pytest.mark.parametrize("seg_len,split_min", [(64, 64), (128, 200), (64, 500)])
@pytest.mark.parametrize("reset", [False, True])
def test_segments_equal_the_unsplit_kernels(fx, seg_len, split_min, reset):
    """A dense 60 k scene on 320 x 240 (lists of up to a few thousand entries): every tile above ``split_min`` entries is
    walked in segments of ``seg_len``.  ``reset``: every opacity at most 0.01 (nothing saturates: no re-walks, every
    tile walks its whole list); otherwise most pixels finish somewhere inside a segment."""
    fused, lib = fx
    from easygaussiansplatting_amd import _lib
    from easygaussiansplatting_amd.function import Camera
    W, H = 320, 240
    sc = S.small_scene(60_000, W, H, 12, seed=5)
    sc.scales[:] = sc.scales * 2.2
    if reset:
        sc.alphas[:] = np.minimum(sc.alphas, 0.01)
    dl = dev(S.normal(3, 21, (3, H, W)).astype(np.float32) / (3 * H * W))
    fused.SEGMENTS = "0"
    ref = run(fused, sc, Camera.from_scene(sc.cam), dl)
    assert not ref["seg"]
    lens = ref["ranges"][:, 1] - ref["ranges"][:, 0]
    assert lens.max() > 4 * seg_len and (lens > split_min).sum() > 20, (lens.max(), (lens > split_min).sum())
    fused.SEGMENTS = "1"
    _lib.check(lib.egs_seg_config(seg_len, split_min, None))
    keep = fused.SEG_SPECULATE
    try:
        # first sight: segment 0 + COMPOSE walking on alone (the backward pass is split regardless) / first sight with
        # every segment of the list speculated (EGS_DRAW_SEG_SPECULATE: most of them wasted on the opaque scene, none on
        # the reset one -- exact either way) / with the walks of earlier renders on record (SPEC items follow them)
        for renders, spec in ((1, "0"), (1, "1"), (3, "auto")):
            fused.SEG_SPECULATE = spec
            got = run(fused, sc, Camera.from_scene(sc.cam), dl, renders)
            assert got["seg"]
            compare(got, ref, "seg%d/%d/%s/r%d/spec%s" % (seg_len, split_min, "reset" if reset else "opaque", renders, spec))
    finally:
        fused.SEG_SPECULATE = keep


def test_segments_skewed_scene_full_size(fx):
    """scene.skewed_scene right after reset_alpha (1.5 M Gaussians, 1080p, lists up to ~13 600 entries, the longest
    walk > 8 000) at the production setting: segment path == unsplit kernels over the whole image and all five
    parameter gradients, and the oracle's blend on four tiles incl. the longest list."""
    fused, lib = fx
    from easygaussiansplatting_amd.function import Camera
    sc = S.skewed_scene(reset_alpha=True)
    W, H = sc.cam.width, sc.cam.height
    dl = dev(S.normal(3, 22, (3, H, W)).astype(np.float32) / (3 * H * W))
    fused.SEGMENTS = "0"
    ref = run(fused, sc, Camera.from_scene(sc.cam), dl)
    fused.SEGMENTS = "auto"
    got = run(fused, sc, Camera.from_scene(sc.cam), dl, 2)
    assert got["seg"]
    lens = ref["ranges"][:, 1] - ref["ranges"][:, 0]
    assert lens.max() > 10_000 and ref["contrib"].max() > 6_000
    # (default tolerance since round 6: tests/test_gpu_round5_vs_oracle.py measures BOTH kernels against the oracle on this
    # scene -- unsplit 1.6e-5, segments 1.7e-4 of the maximum in dL/du, medians 7e-6 / 7e-7 -- so 4e-4 is not needed)
    compare(got, ref, "skewed_reset", flips=64)
    # the oracle on the longest tile, its right neighbour and two others (float64 2D Gaussians of the oracle's own)
    from tests.test_gpu_parity import _oracle_2d
    o_us, o_ci, o_col, _, _ = _oracle_2d(sc, sc.cam)
    tl = int(np.argmax(lens))
    sel = np.array([tl, tl + 1, 0, lens.size // 2], np.int64)
    o_img, o_cont, o_tau = O.draw(W, H, got["ranges"], got["ids"], o_us, o_ci, sc.alphas.astype(np.float64), o_col, None,
                                  O.POLICY_G, tiles=sel)
    gx = (W + 15) // 16
    for t in sel:
        ty, tx = divmod(int(t), gx)
        ys = slice(ty * 16, min(ty * 16 + 16, H)); xs = slice(tx * 16, tx * 16 + 16)
        d = np.abs(got["image"][:, ys, xs] - o_img[:, ys, xs]).max(0)
        flip = got["contrib"][ys, xs] != o_cont[ys, xs]
        assert flip.sum() <= 8 and d[~flip].max() < 1e-4, (t, int(flip.sum()), d.max())


@pytest.mark.parametrize("reset", [False, True])
@pytest.mark.parametrize("how", ["public", "handle"])
def test_seven_op_surface_splits_long_lists(fx, reset, how):
    """``gsplatcu.splat`` / ``splatB`` (ext.cpp:10-32) on the segment path: the public pair -- ``splatB`` is handed
    tensors, so it REBUILDS the segment states from ``contrib`` (egs_splat_bwd_seg, rebuild) -- and the records handle
    (the forward's states are reused).  Lists bit-exact, image / contrib / final_tau and the four gradient tensors equal
    to the unsplit kernels', speculated forward segments included."""
    fused, lib = fx
    from easygaussiansplatting_amd import _lib, gsplatcu as gsc
    W, H = 320, 240
    sc = S.small_scene(60_000, W, H, 3, seed=5)
    sc.scales[:] = sc.scales * 2.2
    if reset:
        sc.alphas[:] = np.minimum(sc.alphas, 0.01)
    cam = sc.cam
    pws, rots, scales, alphas, shs = map(dev, (sc.pws, sc.rots, sc.scales, sc.alphas, sc.shs))
    Rcw, tcw, twc = dev(cam.Rcw), dev(cam.tcw), dev(cam.twc)
    us, pcs, depths = gsc.project(pws, Rcw, tcw, cam.fx, cam.fy, cam.cx, cam.cy, False)
    cov3 = gsc.computeCov3D(rots, scales, depths, False)[0]
    cov2 = gsc.computeCov2D(cov3, pcs, Rcw, depths, cam.fx, cam.fy, cam.width, cam.height, False)[0]
    col = gsc.sh2Color(shs, pws, twc, False)[0]
    cinv, areas = gsc.inverseCov2D(cov2, depths, False)
    dl = dev(S.normal(3, 23, (3, H, W)).astype(np.float32) / (3 * H * W))

    def both():
        d, a = depths.clone(), areas.clone()
        if how == "handle":
            out, h = gsc.splat_with_records(H, W, us, cinv, alphas, d, col, a)
        else:
            out, h = gsc.splat(H, W, us, cinv, alphas, d, col, a), None
        g = gsc.splatB(H, W, us, cinv, alphas, d, col, out[1], out[2], out[3], out[4], dl, records=h)
        return [host(x) for x in out], [host(x) for x in g], h
    keep_spec = fused.SEG_SPECULATE
    try:
        fused.SEGMENTS = "0"
        ref_out, ref_g, _ = both()
        lens = ref_out[3][:, 1] - ref_out[3][:, 0]
        assert lens.max() > 512
        fused.SEGMENTS = "1"
        _lib.check(lib.egs_seg_config(64, 128, None))
        for spec in ("0", "1"):
            fused.SEG_SPECULATE = spec
            out, g, h = both()
            assert how == "public" or (h is not None and h.seg is not None)
            assert np.array_equal(out[3], ref_out[3]) and np.array_equal(out[4], ref_out[4])     # ranges, gsid_per_patch
            flip = out[1] != ref_out[1]
            assert flip.sum() <= 8, int(flip.sum())
            d = np.abs(out[0] - ref_out[0]).max(0)
            assert d[~flip].max() < 2e-5 and np.abs(out[2] - ref_out[2])[~flip].max() < 1e-5
            for a, b, nm in zip(g, ref_g, ("dus", "dcinv", "dalpha", "dcolor")):
                assert_grad_close(a, b, "ops_%s_%s_spec%s:%s" % (how, "reset" if reset else "opaque", spec, nm),
                                  tol_max=1e-4, med_rel=2e-5, max_rel=2e-3, outliers=8)
    finally:
        fused.SEG_SPECULATE = keep_spec


def test_trainer_on_the_segment_path_follows_the_unsplit_trainer(fx):
    """``Trainer`` (deferred validation, in-kernel accumulation over views, factored SH gradient, FusedAdam) with every
    render forced through the segment path (64-entry segments) -- across a densification, which changes N (new problem
    size, no walk on record), and a ``reset_alpha`` (nothing saturates any more) -- follows the trainer on the unsplit
    kernels: same losses, same parameters up to the rounding of float atomics."""
    fused, lib = fx
    from easygaussiansplatting_amd import _lib
    from easygaussiansplatting_amd.function import Camera, render
    from easygaussiansplatting_amd.trainer import Trainer
    W, H = 160, 96
    sc = S.small_scene(20_000, W, H, 12, seed=9)
    sc.scales[:] = sc.scales * 2.0
    cams = [Camera.from_scene(c) for c in S.ring_cameras(sc.cam, 3, radius=5.0)]
    with torch.no_grad():
        gts = [render(dev(sc.pws), dev(sc.shs), dev(sc.alphas), dev(sc.scales), dev(sc.rots), c)[0] for c in cams]
    start = S.small_scene(20_000, W, H, 12, seed=9)
    start.scales[:] = start.scales * 2.0
    start.shs[:, :3] += 0.5 * S.normal(5, 3, (20_000, 3)).astype(np.float32)

    def train(seg):
        fused.SEGMENTS = seg
        tr = Trainer(start, cams, gts, max_steps=100, scene_size=4.0, seed=3)
        losses = [tr.step([0, 1, 2]) for _ in range(4)]
        tr.densify()
        losses += [tr.step([0, 1, 2]) for _ in range(3)]
        tr.reset_alpha()
        losses += [tr.step([2, 0, 1]) for _ in range(4)]
        return losses, {k: host(v) for k, v in tr.params.items()}, tr.redone_steps
    _lib.check(lib.egs_seg_config(64, 64, None))
    l0, p0, _ = train("0")
    l1, p1, _ = train("1")
    assert all(np.isfinite(l1)) and p0["pws"].shape == p1["pws"].shape and p0["pws"].shape[0] != 20_000
    np.testing.assert_allclose(l1, l0, rtol=2e-4, atol=1e-6)
    for k in p0:
        d = np.abs(p1[k] - p0[k])
        # Adam normalises every gradient to ~lr: a gradient that is rounding-level noise around zero may step the other
        # way, so single entries differ by a few lr; the bulk does not
        assert np.median(d) <= 1e-6 + 1e-5 * np.abs(p0[k]).max() and (d > 5e-3 * max(1.0, np.abs(p0[k]).max())).mean() < 1e-3, k


def _segment_states(seg_bytes, T):
    """what k_seg_plan / k_draw_seg left in a segment workspace (mirrors seg_carve, csrc/egs_segments.hip: header,
    seg_base[T], walk[T], items3, two scratch arrays, items1, then per slot [256] x (float4, float, float)) ->
    hdr, seg_base, st4[slot, 256, 4]; a slot's 256 entries are (lane + 64 k): pixel (8 (k & 1) + (lane & 7),
    8 (k >> 1) + (lane >> 3)) of the tile"""
    w = seg_bytes.view(np.int32)
    nbytes = seg_bytes.size
    Tp = (T + 63) // 64 * 64
    fixed = 16 + 48 + 4096 + 5 * Tp                  # header, spare words, the plan's global bins, five [T] arrays
    slots = (nbytes - (4 * (fixed + T + 64) + 1024)) // (256 * 6 * 4 + 4)
    item_cap = min(T + slots, 1 << 20)
    slot_cap = min(slots, item_cap - T)
    hdr = w[:16]
    seg_base = w[64 + 4096:64 + 4096 + T]
    off = 4 * (fixed + T + slots + 64)
    off = (off + 255) // 256 * 256
    st4 = seg_bytes[off:off + slot_cap * 256 * 16].view(np.float32).reshape(slot_cap, 256, 4)
    return hdr, seg_base, st4


def _slot_to_tile(a):
    """[256, ...] in slot order -> [16, 16, ...] (y, x) of the tile"""
    out = np.zeros((16, 16) + a.shape[1:], a.dtype)
    for k in range(4):
        blk = a[64 * k:64 * k + 64].reshape((8, 8) + a.shape[1:])          # [lane >> 3, lane & 7]
        out[8 * (k >> 1):8 * (k >> 1) + 8, 8 * (k & 1):8 * (k & 1) + 8] = blk
    return out


@pytest.mark.parametrize("reset,spec", [(False, "1"), (True, "1"), (True, "0")])
def test_segment_end_states_equal_the_oracles(fx, reset, spec):
    """The states the forward launches leave for the backward pass -- per segment and pixel the transmittance at the
    segment's end and G, the colour of everything behind it -- against oracle/segment_oracle.py (float64, the stages
    evaluated in the device's float32; tests/test_segment_oracle.py shows that decomposition to be the reference's
    loop).  ``spec`` "1": every segment speculated, then fixed and composed; "0": first sight, the composing wave walks
    everything behind segment 0 itself."""
    fused, lib = fx
    from easygaussiansplatting_amd import _lib
    from easygaussiansplatting_amd.function import Camera
    from oracle import segment_oracle as SO
    from tests.test_gpu_parity import _oracle_2d
    W, H, L = 160, 128, 64
    sc = S.small_scene(16_000, W, H, 12, seed=9)
    sc.scales[:] = sc.scales * 2.2
    if reset:
        sc.alphas[:] = np.minimum(sc.alphas, 0.01)
    fused.SEGMENTS = "1"
    _lib.check(lib.egs_seg_config(L, L, None))
    keep = fused.SEG_SPECULATE
    fused.SEG_SPECULATE = spec
    try:
        with torch.no_grad():
            _, _, st = fused.forward(dev(sc.pws), dev(sc.shs), dev(sc.alphas).reshape(-1, 1), dev(sc.scales),
                                     dev(sc.rots), Camera.from_scene(sc.cam), need_grad=True)
            torch.cuda.synchronize()
    finally:
        fused.SEG_SPECULATE = keep
    assert st.seg is not None
    T = ((W + 15) // 16) * ((H + 15) // 16)
    hdr, seg_base, st4 = _segment_states(host(st.seg), T)
    ranges, ids = host(st.ranges), host(st.gaussian_ids())
    lens = ranges[:, 1] - ranges[:, 0]
    split = np.nonzero(seg_base >= 0)[0]
    assert len(split) > 10 and np.array_equal(np.sort(split), np.nonzero(lens > L)[0]) and hdr[6] == L
    tiles = [int(np.argmax(lens))] + [int(t) for t in split[:: max(1, len(split) // 5)][:5]]
    o_us, o_ci, o_col, _, _ = _oracle_2d(sc, sc.cam, dtype=np.float32)
    _, cont, _, states = SO.draw_segments(W, H, ranges, ids, o_us, o_ci, sc.alphas.astype(np.float64), o_col, L,
                                          tiles=tiles)
    assert np.array_equal(cont[cont > 0] > 0, np.ones((cont > 0).sum(), bool))
    n_cmp, n_bad = 0, 0
    worst = 0.0
    for t in tiles:
        G, T_end = states[t]
        y0, x0 = (t // ((W + 15) // 16)) * 16, (t % ((W + 15) // 16)) * 16
        hh, ww = min(16, H - y0), min(16, W - x0)
        for s in range(G.shape[0]):
            alive = T_end[s] > 0                      # the pixel was alive in segment s: the device wrote both states
            if not alive.any():
                break
            d = _slot_to_tile(st4[seg_base[t] + s])[:hh, :ww]
            e_t = np.abs(d[..., 3] - T_end[s])[alive]
            e_g = np.abs(d[..., :3] - np.moveaxis(G[s], 0, -1))[alive]
            n_cmp += e_t.size + e_g.size
            n_bad += int((e_t > 2e-5).sum() + (e_g > 1e-4).sum())
            worst = max(worst, float(e_t.max()), float(e_g.max()))
    assert n_cmp > 20_000
    # (a pixel whose alpha' sits at the skip threshold, or whose tau crosses the stop one entry apart, differs in a
    # whole state: counted, as everywhere in the suite)
    assert n_bad <= 2e-3 * n_cmp, (n_bad, n_cmp, worst)


def test_hint_words_are_a_completed_renders_pair_and_steer_the_path(fx):
    """What the host steers the path by (fused._seg_decision): {longest list, longest walk} of ONE completed render,
    published to the page-locked slot by the NEXT render's range kernel (csrc/egs_bin.hip k_tile_ranges; round 6 found the
    earlier forms -- running maxima, words from two different cameras -- flipping the path mid-epoch).  Two cameras of
    very different walks alternate; after every render the slot must hold exactly the previous render's pair, computed
    here from that render's own ``ranges`` / ``contrib``; long walks select the segment kernels, short ones the unsplit
    kernels, a problem size nothing is known about the segment kernels; ``expect_long_walks`` overrides a short record."""
    fused, lib = fx
    from easygaussiansplatting_amd import _lib
    from easygaussiansplatting_amd.function import Camera
    W, H = 320, 240
    sc = S.small_scene(60_011, W, H, 3, seed=7)          # (a size of its own: no other test has left a hint for it)
    sc.scales[:] = sc.scales * 2.2
    lo = np.minimum(sc.alphas, 0.01).astype(np.float32)   # nothing saturates: every tile walks its whole list
    hi = sc.alphas.astype(np.float32)                     # opaque: a few hundred entries, then the tau stop
    cam_a = Camera.from_scene(sc.cam)
    cam_b = Camera.from_scene(S.ring_cameras(sc.cam, 8)[3])
    key = (sc.n, W, H)
    d = torch.device("cuda", torch.cuda.current_device())
    P = [dev(sc.pws), dev(sc.shs), None, dev(sc.scales), dev(sc.rots)]

    def render(alphas, cam):
        P[2] = dev(alphas).reshape(-1, 1)
        with torch.no_grad():
            _, _, st = fused.forward(*P, cam, need_grad=False)
        torch.cuda.synchronize()
        rg, ct = host(st.ranges), host(st.contrib)
        return (int((rg[:, 1] - rg[:, 0]).max()), int(ct.max())), st.seg is not None

    # the split threshold between the two regimes' walks, measured on a copy of the scene with one Gaussian less (a
    # problem size of its own: the size under test stays unknown to the host)
    keep_n = [t[:-1].contiguous() for t in (P[0], P[1], P[3], P[4])]
    Pn = P
    P = [keep_n[0], keep_n[1], None, keep_n[2], keep_n[3]]
    fused.SEGMENTS = "0"
    lo, hi = lo[:-1], hi[:-1]
    w_lo = min(render(lo, cam_a)[0][1], render(lo, cam_b)[0][1])
    w_hi = max(render(hi, cam_a)[0][1], render(hi, cam_b)[0][1])
    assert w_hi + 64 < w_lo, (w_hi, w_lo)
    split = (w_hi + w_lo) // 2
    P = Pn
    lo = np.minimum(sc.alphas, 0.01).astype(np.float32)
    hi = sc.alphas.astype(np.float32)
    fused.SEGMENTS = "auto"
    _lib.check(lib.egs_seg_config(64, split, None))
    assert fused.seg_hint(d, key) is None
    pair0, seg0 = render(lo, cam_a)
    assert seg0                                           # nothing known about this size: the segment kernels
    assert pair0[1] > split
    seen = [pair0]
    path = []
    for alphas, cam in ((hi, cam_b), (hi, cam_a), (lo, cam_b), (lo, cam_a), (hi, cam_b), (hi, cam_b), (hi, cam_b)):
        pair, seg = render(alphas, cam)
        # published by THIS render's range kernel: the previous render's words, both of them, nothing of this one's
        assert fused.seg_hint(d, key) == seen[-1], (fused.seg_hint(d, key), seen)
        seen.append(pair)
        path.append(seg)
    assert max(p[1] for p in seen[1:3]) < split < min(p[1] for p in seen[3:5]), seen     # (the scene does what it is for)
    # a render decides by what was on record when it was enqueued = the render before the previous one:
    #   render 1 (hi) sees nothing published yet -> unknown -> segments; render 2 sees render 0 (long) -> segments;
    #   render 3 sees render 1 (short) -> unsplit; 4 sees 2 (short) -> unsplit; 5 sees 3 (long) -> segments;
    #   6 sees 4 (long) -> segments; 7 sees 5 (short) -> unsplit
    assert path == [True, True, False, False, True, True, False], (path, seen)
    # the caller that KNOWS (DensityControl.reset_alpha): the next renders take the segment kernels whatever is on record
    assert fused.seg_hint(d, key)[1] < split
    fused.expect_long_walks(d, renders=2)
    assert render(lo, cam_a)[1] and render(lo, cam_a)[1]


def test_segment_path_on_several_streams_equals_one_stream(fx):
    """Four ring views of ``scene.skewed_scene`` right after ``reset_alpha`` (the segment kernels, their per-(size,
    stream) walk words, one hint slot shared by all lanes, a segment workspace per view in flight) as ONE step on three
    HIP streams with deferred validation (``dist_views.ViewStreams``: what ``bench.py --views-per-rank`` and a trainer
    with several views per step run) against the same views one after the other on one stream: same gradient sums
    (float atomics in another order), for three steps in a row (first sight, then with the walks on record)."""
    fused, lib = fx
    from easygaussiansplatting_amd import dist_views as DV
    from easygaussiansplatting_amd.function import Camera, GSFunction
    GSFunction.mode = "fused"
    sc = S.skewed_scene(reset_alpha=True)
    W, H = sc.cam.width, sc.cam.height
    V = 4
    cams = [Camera.from_scene(c) for c in S.ring_cameras(sc.cam, 8)[:V]]
    dls = [dev(S.normal(5, 30 + v, (3, H, W)).astype(np.float32) / (3 * H * W)) for v in range(V)]
    names = ("pws", "shs", "alphas", "scales", "rots")
    P = dict(pws=dev(sc.pws), shs=dev(sc.shs), alphas=dev(sc.alphas).reshape(-1, 1), scales=dev(sc.scales), rots=dev(sc.rots))
    fused.SEGMENTS = "auto"
    fused.expect_long_walks(torch.device("cuda", torch.cuda.current_device()), renders=2 * V)
    # one stream, one view after the other, autograd accumulates
    leaves = [P[k].detach().requires_grad_(True) for k in names]
    us0 = torch.zeros((sc.n, 2), device="cuda", requires_grad=True)
    took_segments = 0
    for v in range(V):
        image, _ = GSFunction.apply(*leaves, us0, cams[v])
        image.backward(dls[v])
    torch.cuda.synchronize()
    ref = [t.grad.clone() for t in leaves]
    assert all(torch.isfinite(r).all() for r in ref)
    # three streams
    vleaves = [P[k].detach().requires_grad_(True) for k in names]
    vs = DV.ViewStreams(vleaves, 3)
    uss = [torch.zeros((sc.n, 2), device="cuda", requires_grad=True) for _ in range(3)]
    for rep in range(3):
        for t in vleaves:
            t.grad = None
        with torch.no_grad(), fused.deferred() as d0:      # which path the views of this step take (forward only)
            st = fused.forward(*[t.detach() for t in vleaves], cams[rep % V])[2]
            d0.commit()
        took_segments += int(st.seg is not None)
        redo = True
        while redo:
            for t in vleaves:
                t.grad = None
            with fused.deferred() as d:
                vs.begin()
                with fused.accumulate_in_kernel():
                    for v in range(V):
                        with vs.lane(v) as lv:
                            image, _ = GSFunction.apply(*lv, uss[vs.lane_index(v)], cams[v])
                            image.backward(dls[v])
                vs.finish()
                redo = bool(d.commit())                      # (a view outgrew the enqueue-ahead buffers: the step again)
        torch.cuda.synchronize()
        for k, t, r in zip(names, vleaves, ref):
            assert float((t.grad - r).abs().max()) <= 2e-5 * float(r.abs().max()), (rep, k)
    assert took_segments == 3
