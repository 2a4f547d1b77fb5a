"""``splat`` -> ``splatB`` record reuse (easygaussiansplatting_amd/gsplatcu.py, ``SplatRecords``) can never hand
``splatB`` records that no longer describe its input tensors.

Ground truth of every case: ``splatB`` on CLONES of the (mutated) tensors -- fresh storage that no handle or memo entry
can match, i.e. the pack-from-scratch path ``egs_splat_bwd`` (the reference's behaviour, gausplat.cu:114-159).

* the PUBLIC pair keeps the forward draw's masked list, order buffer and cleared gradient records and validates them
  by CONTENT on the device (stamps of the values, egs_pack_records_validate): a write through ``.data`` (invisible to
  ``_version``), an in-place op, a policy swap, another stream, other tensors between the two calls -- gradients equal
  the ground truth every time; with ``set_memo(False)`` nothing is kept at all;
* the explicit handle (``splat_with_records`` / ``records=``, what GSFunction mode "ops" uses): reused when nothing
  changed, dropped when a tensor's version moved;
* ``torch.inference_mode()``: ``splat`` works (inference tensors have no version counter; round 3 raised);
* ``fused.accumulate_in_kernel``: ``torch.autograd.grad`` inside the block returns tensors and leaves ``.grad`` alone;
* the alignment words of a flat gradient buffer are zero."""
import numpy as np
import pytest
import torch

from easygaussiansplatting_amd import scene as S

pytestmark = pytest.mark.gpu


@pytest.fixture()
def gsc():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from easygaussiansplatting_amd import gsplatcu
    gsplatcu.set_policy("gsplatcu")
    gsplatcu.set_memo(False)
    yield gsplatcu
    gsplatcu.set_memo(False)
    gsplatcu.set_policy("gsplatcu")


def _inputs(gsc, n=3000, w=160, h=96, seed=7, mod=None):
    sc = S.small_scene(n, w, h, 3, seed=seed)
    if mod == "giants":                   # rects far beyond 4 x 4 tiles, some covering the image
        sc.scales[:40] *= 30.0
        sc.alphas[:40] = np.minimum(sc.alphas[:40], 0.2)
    elif mod == "needles":                # large rects, thin footprints: most tiles of a rect get an empty mask
        sc.scales[:, 0] = 0.4
        sc.scales[:, 1:] = 0.004
    elif mod == "faint":                  # alpha below / barely above alpha_skip: empty and one-pixel footprints
        sc.alphas[:1000] = 0.0015
        sc.alphas[1000:2000] = 0.00205
    cam = sc.cam
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    pws, rots, scales, alphas, shs = map(dev, (sc.pws, sc.rots, sc.scales, sc.alphas, sc.shs))
    Rcw, tcw, twc = dev(cam.Rcw), dev(cam.tcw), dev(cam.twc)
    us, pcs, depths = gsc.project(pws, Rcw, tcw, cam.fx, cam.fy, cam.cx, cam.cy, False)
    cov3 = gsc.computeCov3D(rots, scales, depths, False)[0]
    cov2 = gsc.computeCov2D(cov3, pcs, Rcw, depths, cam.fx, cam.fy, cam.width, cam.height, False)[0]
    col = gsc.sh2Color(shs, pws, twc, False)[0]
    cinv, areas = gsc.inverseCov2D(cov2, depths, False)
    dl = dev(S.normal(3, 1, (3, h, w)))
    return dict(H=h, W=w, us=us, cinv=cinv, alphas=alphas, depths=depths, col=col, areas=areas, dl=dl)


def _truth(gsc, d, out):
    """splatB on clones: nothing kept anywhere can match their storage."""
    c = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in d.items()}
    g = gsc.splatB(c["H"], c["W"], c["us"], c["cinv"], c["alphas"], c["depths"], c["col"], out[1].clone(),
                   out[2].clone(), out[3].clone(), out[4].clone(), c["dl"])
    torch.cuda.synchronize()
    return [x.clone() for x in g]


def _same(a, b):
    for x, y in zip(a, b):
        scale = float(y.abs().max())
        assert scale > 0
        assert float((x - y).abs().max()) <= 2e-5 * scale        # order of the float atomics only


def _splatB(gsc, d, out, **kw):
    return gsc.splatB(d["H"], d["W"], d["us"], d["cinv"], d["alphas"], d["depths"], d["col"], out[1], out[2], out[3],
                      out[4], d["dl"], **kw)


MUTATIONS = ["none", "data_write", "inplace", "policy", "stream", "other_tensor"]


@pytest.mark.parametrize("memo", [False, True])
@pytest.mark.parametrize("what", MUTATIONS)
def test_public_pair_never_differentiates_stale_records(gsc, what, memo):
    gsc.set_memo(memo)
    d = _inputs(gsc)
    out = gsc.splat(d["H"], d["W"], d["us"], d["cinv"], d["alphas"], d["depths"], d["col"], d["areas"])
    stream = None
    if what == "data_write":
        d["us"].data.add_(0.75)                     # _version stays 0
        assert d["us"]._version == 0
    elif what == "inplace":
        d["cinv"].mul_(1.1)
    elif what == "policy":
        gsc.set_policy("forward_cpu"); gsc.set_policy("gsplatcu")      # same policy again: still fine either way
        d["alphas"].mul_(0.9)
    elif what == "stream":
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
    elif what == "other_tensor":
        d["col"] = d["col"] * 0.5 + 0.1
    if stream is not None:
        with torch.cuda.stream(stream):
            got = _splatB(gsc, d, out)
        stream.synchronize()
    else:
        got = _splatB(gsc, d, out)
    torch.cuda.synchronize()
    _same(got, _truth(gsc, d, out))
    if what != "none":                              # and the mutation really changes the answer
        d0 = _inputs(gsc)
        ref0 = _truth(gsc, d0, out)
        assert what == "stream" or any(float((a - b).abs().max()) > 1e-3 * float(b.abs().max())
                                       for a, b in zip(got, ref0))


def test_policy_swap_between_the_calls(gsc):
    """Records packed under one policy are not used under another (forward_cpu needs areas=: checked separately)."""
    d = _inputs(gsc)
    out, h = gsc.splat_with_records(d["H"], d["W"], d["us"], d["cinv"], d["alphas"], d["depths"], d["col"], d["areas"])
    gsc.set_policy("forward_cpu")
    assert not h.matches(d["us"].device, gsc._stream(), (d["us"], d["cinv"], d["alphas"], d["col"]), d["W"], d["H"])
    gsc.set_policy("gsplatcu")
    assert h.matches(d["us"].device, gsc._stream(), (d["us"], d["cinv"], d["alphas"], d["col"]), d["W"], d["H"])


def test_public_pair_validates_by_content(gsc):
    """What the content validation does to the kept list: untouched when splatB gets the values splat saw (the masks are
    used: some are narrower than 0xF), untouched for OTHER tensors holding the same values, repaired entry by entry
    where values changed -- a write through ``.data`` to the first 300 Gaussians turns exactly the entries of the stamp
    blocks those rows live in (rows 0..511) into (caller's index | all four blocks)."""
    gsc.set_memo(True)
    d = _inputs(gsc, n=3000)
    key = (0, int(torch.cuda.current_stream().cuda_stream or 0))
    out = gsc.splat(d["H"], d["W"], d["us"], d["cinv"], d["alphas"], d["depths"], d["col"], d["areas"])
    h = gsc._splat_memo[key]
    assert h.tensors is None and h.rec is None and h.stamp is not None          # no tensor is referenced
    P = out[4].shape[0]
    kept0 = h.lists[:P].clone()
    assert torch.equal(kept0 & 0x0FFFFFFF, out[4]) and int((((kept0 >> 28) & 0xF) != 0xF).sum()) > 0.2 * P
    got = _splatB(gsc, d, out)
    torch.cuda.synchronize()
    assert torch.equal(h.lists[:P], kept0) and h.gpack is None                  # nothing repaired, records handed out
    _same(got, _truth(gsc, d, out))
    # other tensors, same values: still valid (nothing about identity is compared)
    out = gsc.splat(d["H"], d["W"], d["us"], d["cinv"], d["alphas"], d["depths"], d["col"], d["areas"])
    h = gsc._splat_memo[key]
    c = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in d.items()}
    got = gsc.splatB(c["H"], c["W"], c["us"], c["cinv"], c["alphas"], c["depths"], c["col"], out[1], out[2], out[3],
                     out[4], c["dl"])
    torch.cuda.synchronize()
    assert torch.equal(h.lists[:P], kept0)
    _same(got, _truth(gsc, d, out))
    # a raw write to some rows: exactly the entries of their stamp blocks lose their masks
    out = gsc.splat(d["H"], d["W"], d["us"], d["cinv"], d["alphas"], d["depths"], d["col"], d["areas"])
    h = gsc._splat_memo[key]
    d["us"].data[:300] += 0.37
    got = _splatB(gsc, d, out)
    torch.cuda.synchronize()
    kept1 = h.lists[:P]
    touched = (out[4] >> 8) < 2                                                  # Gaussians 0..511: stamp blocks 0 and 1
    assert torch.equal(kept1 & 0x0FFFFFFF, out[4])
    assert bool((((kept1[touched] >> 28) & 0xF) == 0xF).all()) and torch.equal(kept1[~touched], kept0[~touched])
    _same(got, _truth(gsc, d, out))
    # a foreign gsid_per_patch (another list of the same length): every entry is the caller's afterwards
    out = gsc.splat(d["H"], d["W"], d["us"], d["cinv"], d["alphas"], d["depths"], d["col"], d["areas"])
    h = gsc._splat_memo[key]
    other = out[4].flip(0).contiguous()
    out_f = list(out); out_f[4] = other
    gsc.splatB(d["H"], d["W"], d["us"], d["cinv"], d["alphas"], d["depths"], d["col"], out[1], out[2], out[3], other,
               d["dl"])
    torch.cuda.synchronize()
    assert torch.equal(h.lists[:P] & 0x0FFFFFFF, other)


def test_explicit_handle_is_reused_and_dropped(gsc):
    d = _inputs(gsc)
    out, h = gsc.splat_with_records(d["H"], d["W"], d["us"], d["cinv"], d["alphas"], d["depths"], d["col"], d["areas"])
    assert h is not None and h.gpack is not None
    got = _splatB(gsc, d, out, records=h)
    assert h.gpack is None                           # the cleared gradient records were handed out (once)
    _same(got, _truth(gsc, d, out))
    got2 = _splatB(gsc, d, out, records=h)           # a second backward (retain_graph): own buffer, same result
    _same(got2, _truth(gsc, d, out))
    d["alphas"].mul_(0.8)                            # version moved: the handle no longer matches
    assert not h.matches(d["us"].device, gsc._stream(), (d["us"], d["cinv"], d["alphas"], d["col"]), d["W"], d["H"])
    _same(_splatB(gsc, d, out, records=h), _truth(gsc, d, out))


@pytest.mark.parametrize("memo", [False, True])
def test_splat_under_inference_mode(gsc, memo):
    gsc.set_memo(memo)
    with torch.inference_mode():
        d = _inputs(gsc)
        out = gsc.splat(d["H"], d["W"], d["us"], d["cinv"], d["alphas"], d["depths"], d["col"], d["areas"])
        out2, h = gsc.splat_with_records(d["H"], d["W"], d["us"], d["cinv"], d["alphas"], d["depths"], d["col"],
                                         d["areas"])
        assert h is None                             # no version counter -> nothing kept
        got = _splatB(gsc, d, out)
    torch.cuda.synchronize()
    assert torch.equal(out[0], out2[0]) and float(out[0].abs().max()) > 0
    assert all(torch.isfinite(g).all() for g in got)


def test_autograd_grad_inside_accumulate_in_kernel_and_zero_padding(gsc):
    from easygaussiansplatting_amd import fused
    from easygaussiansplatting_amd.function import Camera, GSFunction
    GSFunction.mode = "fused"
    n = 2501                                         # not a multiple of four: every slice of the flat buffer is padded
    sc = S.small_scene(n, 128, 96, 12, seed=3)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    L = [dev(a).requires_grad_(True) for a in (sc.pws, sc.shs, sc.alphas.reshape(-1, 1), sc.scales, sc.rots)]
    us = torch.zeros((n, 2), device="cuda", requires_grad=True)
    cam = Camera.from_scene(sc.cam)
    dl = dev(S.normal(2, 2, (3, 96, 128))) / (3 * 96 * 128)
    with fused.accumulate_in_kernel():
        img, _ = GSFunction.apply(*L, us, cam)
        img.backward(dl)                                         # establishes the one-buffer .grad layout
        flat = fused.flat_grad_buffer(L)
        assert flat is not None and torch.isfinite(flat).all()
        used = torch.zeros_like(flat, dtype=torch.bool)
        for t in L:
            off = (t.grad.data_ptr() - flat.data_ptr()) // 4
            used[off:off + t.grad.numel()] = True
        assert int((~used).sum()) > 0 and not flat[~used].any()  # the alignment words are zero, not allocator garbage
        before = [t.grad.clone() for t in L]
        img2, _ = GSFunction.apply(*L, us, cam)
        gs = torch.autograd.grad(img2, L, dl)                    # must RETURN the gradients ...
        assert all(g is not None for g in gs)
        for g, b, t in zip(gs, before, L):
            assert torch.equal(t.grad, b)                        # ... and leave .grad untouched
            assert float((g - b).abs().max()) <= 2e-5 * float(b.abs().max())
        img3, _ = GSFunction.apply(*L, us, cam)
        img3.backward(dl)                                        # .backward(): accumulated in the kernel
        for b, t in zip(before, L):
            assert float((t.grad - 2 * b).abs().max()) <= 4e-5 * float(b.abs().max())


@pytest.mark.parametrize("mod", [None, "giants", "needles", "faint"])
def test_masked_lists_of_the_seven_op_surface_change_no_output(gsc, mod, monkeypatch):
    """Round 4: ``splat`` bins with exact 8x8-block masks in its list values (egs_splat_bin_pack, EGS_DRAW_MASKED_LISTS)
    and returns the stripped list.  Every output is BIT-identical to the unmasked path (MASKED_LISTS = False: the
    reference's lists + the per-entry box test) -- the masks only skip pixels the reference `continue`s on
    (kernel.cu:246) -- and the backward draw over the masked list (handle path) gives the gradients of the plain one."""
    d = _inputs(gsc, mod=mod)
    args = lambda q: (q["H"], q["W"], q["us"], q["cinv"], q["alphas"], q["depths"].clone(), q["col"], q["areas"].clone())
    monkeypatch.setattr(gsc, "MASKED_LISTS", False)
    ref = gsc.splat(*args(d))
    ref2, h0 = gsc.splat_with_records(*args(d))
    assert h0 is not None and h0.lists is None
    monkeypatch.setattr(gsc, "MASKED_LISTS", True)
    for rep in range(3):                       # (exact sequence, then twice enqueue-ahead with the learnt capacity)
        a = args(d)
        out, h = gsc.splat_with_records(*a)
        torch.cuda.synchronize()
        assert h is not None and h.lists is not None
        for x, y in zip(ref, out):
            assert torch.equal(x, y)
        # the walked list = the returned list + a 4-bit mask above bit 28
        walked = h.lists[:out[4].shape[0]]
        assert torch.equal(walked & 0x0FFFFFFF, out[4])
        masks = (walked >> 28) & 0xF
        if mod in ("needles", "faint"):
            assert int((masks == 0).sum()) > 0.05 * masks.numel()      # entries that reach no block of their tile
        q = dict(d); q["depths"], q["areas"] = a[5], a[7]
        plain = _splatB(gsc, q, out)                                    # public path: box test on the plain list
        masked = _splatB(gsc, q, out, records=h)                        # handle: the masked list
        torch.cuda.synchronize()
        _same(masked, plain)
        # a DIFFERENT gsid tensor (a clone) must not pick the masked list up
        assert gsc._walked_lists(h, out[4], out[3]) is not None
        assert gsc._walked_lists(h, out[4].clone(), out[3]) is None and gsc._walked_lists(h, out[4], out[3].clone()) is None
    _same(masked, _truth(gsc, q, out))


def test_two_forwards_then_two_backwards_through_the_public_pair(gsc):
    """A loss over two views: splat(A), splat(B), then splatB for A and for B.  What the stream keeps belongs to B; A's
    backward must not be touched by it -- the validation turns every kept entry that is not A's own into A's entry with
    a full mask (or, when B's list is shorter than A's, nothing kept is used at all).  Both orders, both gradients
    against splatB on clones."""
    gsc.set_memo(True)
    A = _inputs(gsc, seed=7)
    B = _inputs(gsc, seed=8, mod="giants")
    args = lambda q: (q["H"], q["W"], q["us"], q["cinv"], q["alphas"], q["depths"], q["col"], q["areas"])
    for first, second in ((A, B), (B, A)):
        out1 = gsc.splat(*args(first))
        out2 = gsc.splat(*args(second))
        g1 = _splatB(gsc, first, out1)          # the kept state is out2's
        g2 = _splatB(gsc, second, out2)         # ... which the call above may have repaired towards out1's list
        torch.cuda.synchronize()
        _same(g1, _truth(gsc, first, out1))
        _same(g2, _truth(gsc, second, out2))


def test_words_differ_kernel(gsc):
    """egs_words_differ (ABI 9): bitwise, every position, vector and word paths, the <= 3 words behind the last uint4."""
    import ctypes as C
    from easygaussiansplatting_amd import _lib
    lib = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    g = torch.Generator(device="cuda").manual_seed(4)
    for n in (1, 3, 4, 5, 1023, 4099, 1_000_003):
        base = torch.randint(-2**31, 2**31 - 1, (n + 8,), dtype=torch.int32, device="cuda", generator=g)
        for off_a, off_b in ((0, 0), (4, 0), (1, 1), (0, 3)):          # 16-B aligned pairs and misaligned ones
            a = base[off_a:off_a + n]
            b = a.clone() if off_b == 0 else torch.cat([base[:off_b], a])[off_b:]
            assert b.data_ptr() % 16 == (0 if off_b == 0 else 4 * off_b % 16)
            flag = torch.zeros(1, dtype=torch.int32, device="cuda")
            _lib.check(lib.egs_words_differ(p(a), p(b), n, p(flag), st))
            assert int(flag) == 0, (n, off_a, off_b)
            for pos in sorted({0, n // 2, n - 1, max(0, n - 2), min(n - 1, 4 * (n // 4))}):
                c = b.clone() if off_b == 0 else torch.cat([base[:off_b], b])[off_b:]
                c[pos] ^= 1 << (pos % 31)
                flag.zero_()
                _lib.check(lib.egs_words_differ(p(a), p(c), n, p(flag), st))
                assert int(flag) == 1, (n, off_a, off_b, pos)
    # NaNs compare by their bits (a float comparison would call equal NaNs different and -0.0 == 0.0 equal)
    x = torch.tensor([float("nan"), -0.0, 1.0, 2.0], device="cuda")
    y = x.clone()
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(lib.egs_words_differ(p(x), p(y), 4, p(flag), st))
    assert int(flag) == 0
    y[1] = 0.0
    _lib.check(lib.egs_words_differ(p(x), p(y), 4, p(flag), st))
    assert int(flag) == 1
    _lib.check(lib.egs_words_differ(None, None, 0, p(flag), st))          # nothing to compare: untouched
