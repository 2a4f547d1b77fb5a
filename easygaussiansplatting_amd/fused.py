"""Fused training path: what the reference's ``GSFunction``
(gsplat/gsmodel.py:6-93) computes, in three C-ABI calls per step and without the
436 B/Gaussian of Jacobians ever crossing HBM.

* ``forward``  = project + computeCov3D + computeCov2D + sh2Color + inverseCov2D
  in ONE kernel (which also does the binning's getRects and depth keys), then ``splat``'s sort and draw;
  from the second call on the draw stage is enqueued AHEAD of the read-back of the patch count (see below);
* ``backward`` = ``splatB``'s draw pass into packed per-Gaussian gradient records
  + ONE kernel that re-derives the Jacobians in registers and applies
  backward.md eq (3)(4)(5)(7) (gsmodel.py:71-85).

Results equal the seven-op path (same device functions, csrc/egs_gaussian_math.h);
``tests/test_gpu_parity.py`` checks fused == unfused == oracle.
"""
from __future__ import annotations

import collections
import ctypes as C
import os
import threading
import weakref

import torch

from . import _lib
from . import gsplatcu as _gsc
from .dist_views import flat_grad_buffer  # noqa: F401  (re-exported: the buffer is allocated here)
from .gsplatcu import _alphas, _bin_stage, _chk, _lib_on, _pol, _ptr, _stream, _tiles


ENQUEUE_AHEAD = os.environ.get("EGS_ENQUEUE_AHEAD", "1") != "0"   # knob for A/B measurements and tests
MAILBOX_COPY = os.environ.get("EGS_MAILBOX_COPY", "0") == "1"     # A/B knob: read-back by copy instead of kernel stores
REUSE_ORDER = os.environ.get("EGS_BWD_REUSE_ORDER", "1") != "0"   # A/B knob: see KEEP_FORWARD_ORDER
KEEP_FORWARD_ORDER = 16   # include/egs_hip.h EGS_BWD_KEEP_FORWARD_ORDER
ORDER_REFRESH = max(2, int(os.environ.get("EGS_TILE_ORDER_REFRESH", "4")))   # renders of a camera between order refreshes
TILE_WORK_CACHE = os.environ.get("EGS_TILE_WORK_CACHE", "1") != "0"  # A/B knob: forward dispatch order by remembered work
SAVE_DCOLOR = os.environ.get("EGS_SAVE_DCOLOR", "1") != "0"          # A/B knob: forward keeps dcolor/dpw for backward
CULL_LISTS = os.environ.get("EGS_CULL_LISTS", "1") != "0"            # A/B knob: footprint-culled tile lists
# long tile lists split over several waves (include/egs_hip.h egs_splat_draw_rec_seg): "auto" = whenever the longest list
# of the scene's last render exceeded the split threshold (and at first sight), "1" always, "0" never
SEGMENTS = os.environ.get("EGS_SEGMENTS", "auto")
SEG_HISTORY = 4           # include/egs_hip.h EGS_DRAW_SEG_HISTORY
SEG_SPECULATE_FLAG = 8    # include/egs_hip.h EGS_DRAW_SEG_SPECULATE
# a camera without a walk on record on the segment path: "auto" = all its segments at once when the scene's recent renders
# walked at least half of their longest list (nothing saturates: reset_alpha), "1" always, "0" never (segment 0 only)
SEG_SPECULATE = os.environ.get("EGS_SEG_SPECULATE", "auto")
CULLED_LISTS = 32         # include/egs_hip.h EGS_BWD_CULLED_LISTS
ACCUMULATE = 64           # include/egs_hip.h EGS_BWD_ACCUMULATE
FACTORED_SH = 128         # include/egs_hip.h EGS_BWD_FACTORED_SH
GSID_MASK = 0x0FFFFFFF    # csrc/egs_common.h EGS_GSID_MASK
MAILBOX_SLOTS = 64
HINT_SLOTS = 16           # problem sizes that keep a hint slot (longest list / longest walk of their recent renders)


class FusedState:
    """Tensors the backward pass needs (all produced by ``forward``).  ``ticket`` is set while the render's
    patch count has not been validated yet (deferred validation, see ``deferred``)."""
    __slots__ = ("us", "depths", "cinv2ds", "colors", "areas", "rec", "contrib", "final_tau", "ranges", "gsid",
                 "order", "order_by_work", "gpack", "dcw", "culled", "width", "height", "ticket", "_patches", "_keep",
                 "seg")

    def patch_count(self) -> int:
        """P of this render (waits for its read-back if it has not been looked at yet)."""
        if self.ticket is not None:
            _settle(self.ticket, True)
        return self._patches

    def gaussian_ids(self):
        """The tile lists as Gaussian indices, int32[P].  With footprint-culled lists (``culled``) the raw ``gsid``
        values carry the tile's 4-bit block mask above the low 28 bits; this strips it."""
        g = self.gsid[:self.patch_count()]
        return (g & GSID_MASK) if self.culled else g

    def block_masks(self):
        """The 4-bit mask of 8x8 pixel blocks per list entry (bit k = block (k & 1, k >> 1)); 15 for unculled lists."""
        g = self.gsid[:self.patch_count()]
        return ((g >> 28) & 15) if self.culled else torch.full_like(g, 15)


class _Ticket:
    """One enqueue-ahead render whose {P, max depth key} read-back is still in flight."""
    __slots__ = ("ctx", "slot", "key", "cap", "hint", "state", "status", "patches", "need", "collected")
    PENDING, OK, FAILED = 0, 1, 2


class _DeviceCtx:
    """Per-device host state of the fused path: the mailbox, what was learnt about each problem size
    (patch-list capacity, significant depth-key bits) and the renders awaiting validation.  Nothing here
    is shared between devices; access is serialised by ``lock`` (autograd runs backward on its own thread)."""

    def __init__(self, lib, index):
        self.index = index
        self.lib = lib
        self.mb = lib.egs_mailbox_create(MAILBOX_SLOTS)
        if not self.mb:
            raise RuntimeError("egs_mailbox_create failed (page-locked host memory)")
        self.free = list(range(MAILBOX_SLOTS))
        self.pending = collections.deque()
        self.failed = []
        self.capacity = {}      # (N, W, H) -> patch-list allocation size learnt from earlier renders
        # (camera, stream) -> (weakref, its [order | work] buffer, renders so far, problem size): the dispatch order
        # of the tiles is kept between the renders of a camera (a trainer meets every view again each epoch)
        self.tile_work = {}
        self.seg_hint = {}      # (N, W, H) -> mailbox slot kept as the landing zone of "longest list of the last render"
        # ((N, W, H), stream) -> one persistent int32 device word (-1 = nothing gathered): the draw items of a render
        # gather its longest walk there, the next render on that stream publishes it into the hint slot (egs_hip.h)
        self.walk_word = {}
        self.long_walks_expected = 0   # renders for which a caller announced long walks (expect_long_walks)
        self.lock = threading.RLock()


_contexts = {}
_tls = threading.local()
_exchange_hook = None    # dist_views.ChunkedExchange while attached (process-wide: backward runs on autograd's thread)
_sh_sink = None          # dist_views.FactoredShGrad while attached: the SH gradient of a view stays dL/dcolour [N,3]


def _ctx(dev) -> _DeviceCtx:
    c = _contexts.get(dev.index)
    if c is None:
        c = _contexts.setdefault(dev.index, _DeviceCtx(_lib.load(), dev.index))
    return c


def _grow(p):
    return p + p // 16 + 4096


SIZE_TABLE_MAX = 1024    # problem sizes (N, W, H) a process remembers a patch capacity / depth-key hint for


def _learn_capacity(ctx, key, patches):
    """Raise the enqueue-ahead patch capacity of a problem size (ctx.lock held by the caller or not needed: one dict
    store); the table is bounded -- a process that meets ever new sizes (a densifying trainer: one per densification; a
    server rendering many scenes) forgets the sizes it met first."""
    cap = ctx.capacity
    val = max(cap.pop(key, 0), _grow(min(patches, 2**31 - 1)))
    cap[key] = val                                      # (re-inserted: most recently learnt)
    while len(cap) > SIZE_TABLE_MAX:
        cap.pop(next(iter(cap)), None)


def _settle(t: _Ticket, blocking: bool) -> bool:
    """Look at the read-back of one render: True once it has been validated (either way)."""
    ctx = t.ctx
    with ctx.lock:
        if t.status != _Ticket.PENDING:
            return True
        out = (C.c_uint32 * 2)()
    rc = ctx.lib.egs_mailbox_fetch(ctx.mb, t.slot, 1 if blocking else 0, out)   # (the wait holds no lock, no GIL)
    if rc == 0:
        return False
    if rc < 0:
        # the slot never received its values (a HIP error behind the binning stage): the ticket is settled as
        # FAILED and its slot handed back, so that later commit() calls do not trip over it again
        with ctx.lock:
            if t.status == _Ticket.PENDING:
                t.status, t.patches, t.need = _Ticket.FAILED, 0, 0
                ctx.free.append(t.slot)
                try:
                    ctx.pending.remove(t)
                except ValueError:
                    pass
                if t.state is not None:
                    t.state.ticket = None
                    t.state._patches = 0
        _lib.check(-rc)
    with ctx.lock:
        if t.status != _Ticket.PENDING:
            return True
        t.patches, mk = int(out[0]), int(out[1])
        t.need = mk.bit_length()
        ctx.free.append(t.slot)
        try:
            ctx.pending.remove(t)
        except ValueError:
            pass
        ok = t.patches <= t.cap and not (t.hint < 32 and t.need > t.hint) and t.patches < 2**31
        # what the next render of this size starts from
        _gsc._learn_key_bits(ctx.index, t.key, t.need, missed=(t.hint < 32 and t.need > t.hint))
        _learn_capacity(ctx, t.key, t.patches)
        t.status = _Ticket.OK if ok else _Ticket.FAILED
        S = t.state
        if S is not None:
            S._patches = t.patches
            S.ticket = None
            if ok:
                S.gsid = S.gsid[:t.patches]
        if not ok and not t.collected:
            ctx.failed.append(t)
    return True


class deferred:
    """``with fused.deferred() as d: ...; bad = d.commit()`` -- renders inside the block are NOT validated
    when ``forward`` returns: the host never waits for the 8-byte read-back of the patch count inside a
    step and can run a whole step ahead of the GPU.  ``commit()`` validates everything rendered so far
    (it waits for the binning stage of the last render, not for its draw or backward kernels) and returns the
    ``FusedState`` objects whose patch list outgrew the enqueue-ahead capacity or whose depth keys outgrew the
    sort's bit hint: their images and gradients are INCOMPLETE and must be recomputed before anything
    consumes them (the learnt capacity / hint are already raised, so recomputing succeeds).  Leaving the
    block with such a failure uncollected raises."""

    def __enter__(self):
        self._prev = getattr(_tls, "deferred", False)
        _tls.deferred = True
        return self

    def commit(self):
        return commit()

    def __exit__(self, et, ev, tb):
        _tls.deferred = self._prev
        if et is None and not self._prev:
            bad = commit()
            if bad:
                raise RuntimeError("%d enqueue-ahead render(s) were incomplete (patch capacity or depth-key hint "
                                   "exceeded) and nobody collected them with commit(): their results must not be used"
                                   % len(bad))
        return False


def commit(device=None):
    """Validate every render of ``device`` (default: the current one) that is still awaiting its read-back;
    -> list of the FusedState objects that turned out incomplete since the last commit."""
    index = torch.cuda.current_device() if device is None else torch.device(device).index
    ctx = _contexts.get(index)
    if ctx is None:
        return []
    while True:
        with ctx.lock:
            t = ctx.pending[0] if ctx.pending else None
        if t is None:
            break
        _settle(t, True)
    with ctx.lock:
        bad, ctx.failed = ctx.failed, []
    for t in bad:
        t.collected = True
    return [t.state for t in bad]


def expect_long_walks(device=None, renders=4):
    """A caller that KNOWS the next renders will walk their tile lists far (this package's ``DensityControl.reset_alpha``:
    every opacity drops to 0.01, nothing saturates any more, gsmodel.py:320-324) says so: the next ``renders`` renders on
    ``device`` take the segment path and speculate whole lists at once, instead of learning it from the hint words two
    renders late (the draw stage publishes a render's longest walk at the start of the NEXT draw stage on its stream;
    13 + 10 ms instead of 3.3 per training step on scene.skewed_scene's ring views).  An unmodified reference caller never
    calls this and pays those two steps."""
    index = torch.cuda.current_device() if device is None else torch.device(device).index
    if index is None:
        return
    ctx = _contexts.get(index)
    if ctx is None:
        ctx = _ctx(torch.device("cuda", index))
    with ctx.lock:
        ctx.long_walks_expected = max(ctx.long_walks_expected, int(renders))


def _seg_decision(ctx, lib, key, pol_):
    """-> (use the segment path for this render, device-visible address of the hint slot or None).  The draw stage
    leaves two numbers in a page-locked slot kept per problem size -- the longest list, and the longest WALK (largest
    contributor index of a tile) of a recent render -- and a later render looks at them WITHOUT waiting (they may be a
    render or two old; they only select between two exact paths): a scene whose tiles are all walked for less than the
    split threshold takes the unsplit kernels (three launches less), at first sight and from then on long walks take
    the segment path."""
    _tls.seg_speculate = SEG_SPECULATE == "1"
    if SEGMENTS == "0" or pol_.footprint != 0 or not (pol_.alpha_skip > 0) or not (pol_.tau_stop > 0):
        return False, None
    with ctx.lock:
        slot = ctx.seg_hint.pop(key, None)
        if slot is None:
            # at most HINT_SLOTS problem sizes keep a slot; the least recently used one hands its slot on (a kernel of
            # that size still in flight may write into it once more: a stale hint, never a wrong result)
            if len(ctx.seg_hint) >= HINT_SLOTS:
                slot = ctx.seg_hint.pop(next(iter(ctx.seg_hint)))
            elif len(ctx.free) > MAILBOX_SLOTS // 2:     # (never starve the renders of their read-back slots)
                slot = ctx.free.pop()
            else:
                return SEGMENTS == "1", None
            _lib.check(lib.egs_mailbox_clear(ctx.mb, slot))
        ctx.seg_hint[key] = slot                         # (re-inserted: most recently used)
    out = (C.c_uint32 * 4)()
    _lib.check(lib.egs_mailbox_peek(ctx.mb, slot, out))
    cfg = (C.c_int * 2)()
    _lib.check(lib.egs_seg_config(0, 0, cfg))
    longest, walk = int(out[0]), int(out[1])
    known = walk != 0xFFFFFFFF and longest != 0xFFFFFFFF
    # no walk on record yet: the longest LIST bounds it (a scene whose lists all stay below the split threshold never
    # pays for the segment workspace, ~6 KB per 256 entries of a split tile); nothing known at all: the segment path
    unknown = walk == 0xFFFFFFFF and (longest == 0xFFFFFFFF or longest > cfg[1])
    use = SEGMENTS == "1" or unknown or (walk != 0xFFFFFFFF and walk > cfg[1])
    with ctx.lock:
        announced = ctx.long_walks_expected > 0
        if announced:
            ctx.long_walks_expected -= 1
    if announced and SEGMENTS != "0":
        _tls.seg_speculate = SEG_SPECULATE != "0"
        return True, C.c_void_p(lib.egs_mailbox_slot(ctx.mb, slot))
    _tls.seg_speculate = SEG_SPECULATE == "1" or (SEG_SPECULATE == "auto" and known and 2 * walk >= longest)
    return use, C.c_void_p(lib.egs_mailbox_slot(ctx.mb, slot))


def _walk_word(ctx, key, dev, st, have_hint):
    """The persistent device word of (problem size, stream) for the draw stage's longest-walk report, or None."""
    if not have_hint:
        return None
    k = (key, int(st.value or 0))
    with ctx.lock:
        w = ctx.walk_word.get(k)
        if w is None:
            while len(ctx.walk_word) >= 4 * HINT_SLOTS:          # (bounded: streams and sizes that are gone)
                ctx.walk_word.pop(next(iter(ctx.walk_word)))
            w = ctx.walk_word[k] = torch.full((16,), -1, dtype=torch.int32, device=dev)
    return w


def seg_hint(device=None, key=None):
    """(longest list, longest walk) the draw stage last reported for problem size ``key`` = (N, W, H) on ``device``
    (None: nothing yet / no slot) -- what ``_seg_decision`` steers by; for bench lines and tests."""
    index = torch.cuda.current_device() if device is None else torch.device(device).index
    ctx = _contexts.get(index)
    if ctx is None:
        return None
    with ctx.lock:
        slot = ctx.seg_hint.get(key)
    if slot is None:
        return None
    out = (C.c_uint32 * 4)()
    _lib.check(ctx.lib.egs_mailbox_peek(ctx.mb, slot, out))
    f = lambda v: None if v == 0xFFFFFFFF else int(v)
    return f(out[0]), f(out[1])


def _split_sh(low_shs, high_shs, n):
    low = _chk(low_shs, "low_shs", torch.float32, (n, 3))
    high = _chk(high_shs, "high_shs", torch.float32, (n, None))
    K = 3 + high.shape[1]
    if K not in (3, 12, 27, 48):
        raise ValueError("low_shs + high_shs must have 3, 12, 27 or 48 columns, got %d" % K)
    return low, high, K


def forward(pws, shs, alphas, scales, rots, cam, high_shs=None, need_grad=False):
    """-> (image[3,H,W], mask[N] bool, state).  ``cam`` carries Rcw/tcw/twc device
    tensors and fx, fy, cx, cy, width, height (reference gausplat_dataset.py:14-26).
    ``need_grad``: a backward pass will follow (the draw kernel then also zeroes its gradient records).
    With ``high_shs`` the inputs are the RAW training tensors (``shs`` = low_shs, ``alphas`` =
    alphas_raw, ``scales`` = scales_raw, ``rots`` = rots_raw) and the activations of
    gsplat/utils.py:121-150 run inside the kernel (egs_fused_forward_raw)."""
    raw = high_shs is not None
    pws = _chk(pws, "pws", torch.float32, (None, 3))
    n = pws.shape[0]
    if raw:
        shs, high_shs, K = _split_sh(shs, high_shs, n)
    else:
        shs = _chk(shs, "shs", torch.float32, (n, None))
        K = shs.shape[1]
        if K not in (3, 12, 27, 48):
            raise ValueError("shs must have 3, 12, 27 or 48 columns, got %d" % K)
    alphas = _alphas(alphas, n)
    scales = _chk(scales, "scales", torch.float32, (n, 3))
    rots = _chk(rots, "rots", torch.float32, (n, 4))
    Rcw = _chk(cam.Rcw, "cam.Rcw", torch.float32, (3, 3))
    tcw = _chk(cam.tcw, "cam.tcw", torch.float32, (3,))
    twc = _chk(cam.twc, "cam.twc", torch.float32, (3,))
    W, H = int(cam.width), int(cam.height)
    lib = _lib_on(pws)
    dev = pws.device
    pol = C.byref(_pol())
    st = _stream()
    f32, i32 = torch.float32, torch.int32
    S = FusedState()
    S.width, S.height = W, H
    S.ticket, S._patches, S._keep, S.seg = None, None, None, None
    # the draw kernels (forward and backward) work from the packed records alone: us / cinv2ds / colors /
    # areas are not materialised
    S.us = S.cinv2ds = S.colors = S.areas = None
    # Footprint-culled lists: a Gaussian is listed only for the tiles of its rect that {alpha' >= alpha_skip} can
    # reach, and every list value carries the tile's block mask (include/egs_hip.h EGS_DRAW_CULLED_LISTS).  The
    # lists are internal to this path -- the seven-op surface always returns the reference's.
    pol_ = _pol()
    S.culled = bool(CULL_LISTS and pol_.footprint == 0 and pol_.alpha_skip > 0 and n < (1 << 28))
    S.depths = torch.empty((n,), dtype=f32, device=dev)
    S.rec = torch.empty((max(n, 1), 12), dtype=f32, device=dev)   # packed 2D records, reused by backward
    mask = torch.empty((n,), dtype=torch.bool, device=dev)        # depths > 0.2, written by the kernel
    ws_bin_bytes = lib.egs_splat_bin_ws_bytes(n)
    ws_bin = torch.empty(ws_bin_bytes, dtype=torch.uint8, device=dev)
    host_slot = [None]       # mailbox slot the binning kernels also write {P, max key} into (enqueue-ahead path)
    tail = lambda hint, total: (_ptr(alphas), _ptr(Rcw), _ptr(tcw), _ptr(twc), float(cam.fx), float(cam.fy),
                                float(cam.cx), float(cam.cy), W, H, pol, _ptr(S.us), _ptr(S.depths), _ptr(S.cinv2ds),
                                _ptr(S.colors), _ptr(S.areas), _ptr(S.rec), _ptr(mask), _ptr(S.dcw),
                                1 if S.culled else 0, hint, _ptr(ws_bin), ws_bin_bytes, _ptr(total), host_slot[0], st)
    image = torch.empty((3, H, W), dtype=f32, device=dev)       # fully written by the draw stage
    S.contrib = torch.empty((H, W), dtype=i32, device=dev)
    S.final_tau = torch.empty((H, W), dtype=f32, device=dev)
    S.ranges = torch.empty((_tiles(W, H), 2), dtype=i32, device=dev)
    S.order = None            # [tile dispatch order | per-tile work]: ONE buffer per camera, see below
    # packed gradient records of the backward pass: zeroed on the side by the forward draw kernel (one use)
    S.gpack = torch.empty((max(n, 1), 12), dtype=f32, device=dev) if (need_grad and n > 0) else None
    # dcolor/dpw per Gaussian, written by the preprocess kernel for the backward pass: that pass then never reads the
    # SH coefficients (36 B written + read instead of a 4K-byte row re-read; EGS_SAVE_DCOLOR=0: A/B knob)
    S.dcw = torch.empty((n, 9), dtype=f32, device=dev) if (need_grad and n > 0 and SAVE_DCOLOR) else None

    def draw_exact(patches, redo=False):
        # ``redo``: the draw stage of this render ran once already on truncated lists (more patches than the enqueue-ahead
        # buffers held).  Its range kernel has published the PREVIOUS render's hint words; what that truncated draw
        # raised in the walk word is nobody's longest walk: the second range kernel clears it without publishing
        # (hint address withheld) -- otherwise a later render steers by (1189, 696) where the render walked (2063, 696)
        S.gsid = torch.empty(patches, dtype=i32, device=dev)
        ws_draw = torch.empty(lib.egs_splat_draw_ws_bytes(n, patches, W, H), dtype=torch.uint8, device=dev)
        if use_seg:
            S.seg = torch.empty(lib.egs_seg_ws_bytes(max(patches, 1), W, H), dtype=torch.uint8, device=dev)
        # (seg_ws NULL: the unsplit kernels; the hint slot still learns how far this camera's tiles are walked)
        _lib.check(lib.egs_splat_draw_rec_seg(n, patches, None, W, H, _ptr(S.rec), pol, _ptr(ws_bin), _ptr(ws_draw),
                                              ws_draw.numel(), _ptr(image), _ptr(S.contrib), _ptr(S.final_tau),
                                              _ptr(S.ranges), _ptr(S.gsid), _ptr(S.order), _ptr(S.gpack), prev_work,
                                              order_ready, draw_flags, _ptr(S.seg),
                                              S.seg.numel() if S.seg is not None else 0,
                                              None if (redo and walk_word is not None) else seg_hint, _ptr(walk_word), None, st))

    if raw:
        enqueue_bin = lambda hint, total: _lib.check(lib.egs_fused_forward_raw(
            n, K, _ptr(pws), _ptr(rots), _ptr(scales), _ptr(shs), _ptr(high_shs), *tail(hint, total)))
    else:
        enqueue_bin = lambda hint, total: _lib.check(lib.egs_fused_forward(
            n, K, _ptr(pws), _ptr(rots), _ptr(scales), _ptr(shs), *tail(hint, total)))

    ctx = _ctx(dev)
    key = (n, W, H)
    S.ticket = None
    # Dispatch order of the tiles.  A camera keeps ONE [order | work] buffer across its renders (per stream): the
    # draw kernel leaves the work it measured per tile in the second half, and the order in the first half is
    #   first render of the camera   sorted by list length (inside the library),
    #   second render                sorted by the work the first one measured,
    #   later renders                used as it stands -- the work pattern of a camera drifts slowly -- and
    #                                refreshed from the latest work every ORDER_REFRESH-th render:
    # no order kernel at all on most renders (10 us forward, 8 us backward at 1080p).
    prev_work, order_ready = None, 0
    cache_entry = None        # registered only AFTER the draw stage that writes the order buffer was enqueued
    use_seg, seg_hint = _seg_decision(ctx, lib, key, pol_) if n > 0 else (False, None)
    walk_word = _walk_word(ctx, key, dev, st, seg_hint is not None)
    walk_known = False        # the camera was rendered before: its walk lengths are on record
    if TILE_WORK_CACHE and n > 0:
        ck = (id(cam), int(st.value or 0))              # one entry per camera and stream, whatever the scene size
        olen = lib.egs_tile_order_len(W, H)
        with ctx.lock:
            hit = ctx.tile_work.get(ck)
            if hit is not None and hit[0]() is cam and hit[1].numel() == olen and hit[3] == (n, W, H):
                S.order = hit[1]
                # `renders` counts the renders that wrote the ORDER part (the segment path plans its own work items and
                # leaves it alone; work and walk are written by both paths)
                renders = hit[2] + (0 if use_seg else 1)
                walk_known = True
                if not use_seg:
                    if renders <= 2 or renders % ORDER_REFRESH == 0:   # [order | work | walk]: its own work part
                        prev_work = C.c_void_p(S.order.data_ptr() + 4 * (olen - 2 * _tiles(W, H)))
                    else:
                        order_ready = 1
            else:
                S.order = torch.empty(olen, dtype=i32, device=dev)
                renders = 0 if use_seg else 1
        cache_entry = (ck, renders)
    if S.order is None:
        S.order = torch.empty(lib.egs_tile_order_len(W, H), dtype=i32, device=dev)

    def remember_order():
        """The render that just enqueued its draw stage wrote [order | work] (every path of the library does,
        patches == 0 included): only now may the NEXT render of this camera rely on it.  A render that raised
        before this point leaves the cache as it was.  The entry goes when the camera object dies (weakref
        callback): callers that build a Camera per frame do not pile up order buffers."""
        if cache_entry is None:
            return
        ck, renders = cache_entry
        tw = ctx.tile_work

        def drop(_ref, ck=ck, tw=tw, lock=ctx.lock):
            with lock:
                ent = tw.get(ck)
                if ent is not None and ent[0] is _ref:
                    del tw[ck]
        try:
            ref = weakref.ref(cam, drop)
        except TypeError:                               # a camera object that cannot be weakly referenced
            return
        with ctx.lock:
            tw[ck] = (ref, S.order, renders, (n, W, H))
    S.order_by_work = prev_work is not None or order_ready == 1
    draw_flags = (1 if S.culled else 0) | (SEG_HISTORY if (use_seg and walk_known) else 0) | \
        (SEG_SPECULATE_FLAG if (use_seg and getattr(_tls, "seg_speculate", False)) else 0)
    # (SPECULATE with a walk on record: the plan distrusts a record that is far shorter than the tile's list while the
    # scene's recent renders walk most of theirs -- the renders right after reset_alpha, gsmodel.py:320-324)
    cap = ctx.capacity.get(key, 0) if ENQUEUE_AHEAD else 0

    def render_exact(redo=False):
        """Synchronous form: read P back (8 bytes, as the reference does at gausplat.cu:67), then draw."""
        patches = _bin_stage(enqueue_bin, dev, key)
        draw_exact(patches, redo)
        remember_order()
        S._patches = patches
        if n > 0:
            with ctx.lock:
                _learn_capacity(ctx, key, patches)

    if cap == 0 or n == 0:
        render_exact()                               # first render of this size
        return image, mask, S
    # The draw stage is enqueued AHEAD of the read-back: buffers sized by the largest patch count seen so
    # far, the kernels take the real count from device memory, and {P, max depth key} travel to a page-locked
    # mailbox slot by a copy enqueued between the two stages (egs_mailbox_post).  The GPU never waits for the
    # host (the reference, like the seven-op path, idles around cudaMemcpy(&P), gausplat.cu:67).  An overflow
    # of the capacity or of the depth-key hint is detected after the fact -- here (one C-side wait on the
    # slot's event) or, inside a ``deferred()`` block, at ``commit()`` -- and the render is redone.
    t = _Ticket()
    t.ctx, t.key, t.cap, t.state, t.status, t.collected = ctx, key, cap, S, _Ticket.PENDING, False
    t.hint = _gsc._get_key_bits(dev.index, key)
    while True:
        with ctx.lock:
            if ctx.free:
                t.slot = ctx.free.pop()
                break
            oldest = ctx.pending[0] if ctx.pending else None
        if oldest is None:
            raise RuntimeError("fused.forward: no mailbox slot free and no render in flight (slots leaked)")
        _settle(oldest, True)                         # every slot in flight: wait for the oldest render
    try:
        total = torch.empty(2, dtype=i32, device=dev)
        if MAILBOX_COPY:          # {P, max key} by an 8-byte device-to-host copy behind the binning stage
            enqueue_bin(t.hint, total)
            _lib.check(lib.egs_mailbox_post(ctx.mb, t.slot, _ptr(total), st))
        else:                     # the binning kernels store them into the page-locked slot themselves
            _lib.check(lib.egs_mailbox_arm(ctx.mb, t.slot, st))
            host_slot[0] = C.c_void_p(lib.egs_mailbox_slot(ctx.mb, t.slot))
            enqueue_bin(t.hint, total)
            host_slot[0] = None   # (a later synchronous re-render must not write into a slot that was handed back)
        gsid_full = torch.empty(cap, dtype=i32, device=dev)
        ws_draw = torch.empty(lib.egs_splat_draw_ws_bytes(n, cap, W, H), dtype=torch.uint8, device=dev)
        if use_seg:
            S.seg = torch.empty(lib.egs_seg_ws_bytes(cap, W, H), dtype=torch.uint8, device=dev)
        _lib.check(lib.egs_splat_draw_rec_seg(n, cap, _ptr(total), W, H, _ptr(S.rec), pol, _ptr(ws_bin),
                                              _ptr(ws_draw), ws_draw.numel(), _ptr(image), _ptr(S.contrib),
                                              _ptr(S.final_tau), _ptr(S.ranges), _ptr(gsid_full), _ptr(S.order),
                                              _ptr(S.gpack), prev_work, order_ready, draw_flags, _ptr(S.seg),
                                              S.seg.numel() if S.seg is not None else 0, seg_hint, _ptr(walk_word), None, st))
    except BaseException:
        # Whatever was enqueued before the failure (the arm, the binning chain) still stores {P, max key} into the
        # slot: it goes back on the free list only once those kernels have run -- otherwise a render on another
        # ViewStreams lane could pick it up and settle on THEIR values.  Rare path: a stream wait is fine.
        try:
            torch.cuda.current_stream(dev).synchronize()
        except Exception:
            pass
        with ctx.lock:
            t.status = _Ticket.FAILED
            ctx.free.append(t.slot)
        raise
    remember_order()
    S.gsid = gsid_full                                # entries past P are unused (the kernels walk `ranges`)
    S._patches = None
    S.ticket = t
    with ctx.lock:
        ctx.pending.append(t)
    if getattr(_tls, "deferred", False):
        with ctx.lock:                                # look at whatever has landed meanwhile (no waiting)
            waiting = list(ctx.pending)
        for old in waiting:
            if old is not t and not _settle(old, False):
                break
        return image, mask, S
    t.collected = True                                # validated right here: never reported by commit()
    _settle(t, True)
    if t.status == _Ticket.FAILED:
        if t.patches >= 2**31:
            raise RuntimeError("splat: %d tile patches overflow int32 indexing" % t.patches)
        if t.hint < 32 and t.need > t.hint:           # stale depth-key hint: everything again
            render_exact(redo=True)
        else:                                         # more patches than ever before: redo the draw stage
            draw_exact(t.patches, redo=True)
    return image, mask, S


_pad_index = {}    # (device, N, slice widths) -> positions of the alignment words of a flat gradient buffer


class accumulate_in_kernel:
    """``with fused.accumulate_in_kernel(): ...`` -- backward passes of ``GSFunction`` / ``GSRawFunction`` inside the
    block ADD their parameter gradients to the ``.grad`` the leaves already hold, inside the chain-rule kernel, and
    return ``None`` for them to autograd (which then leaves ``.grad`` alone) -- instead of handing autograd fresh
    tensors that it accumulates with separate kernels (976 B per Gaussian and view against 488).  For a rank that
    renders several views per step.  Only taken when every differentiated input is a leaf whose ``.grad`` came out
    of this module's backward (slices of one buffer, ``flat_grad_buffer``); the first view of a step, non-leaf inputs
    or foreign ``.grad`` tensors go the ordinary way.  Tensor hooks / post-accumulate hooks of the leaves do not
    fire for the views accumulated this way."""

    def __enter__(self):
        self._prev = getattr(_acc_flag, "on", False)
        _acc_flag.on = True
        return self

    def __exit__(self, et, ev, tb):
        _acc_flag.on = self._prev
        return False


class _AccFlag:          # process-wide, not thread-local: autograd runs backward on its own thread
    on = False


_acc_flag = _AccFlag()


def _engine_accumulates_into(node_ctx, count):
    """True when the running autograd pass will ACCUMULATE into ``.grad`` of the first ``count`` inputs of the node
    ``node_ctx`` (a ``.backward()`` that reaches all of them); False under ``torch.autograd.grad`` (gradients are
    captured and returned, ``.grad`` must stay untouched) or ``backward(inputs=[...])`` that leaves some out."""
    try:
        nodes = [fn for fn, _ in node_ctx.next_functions[:count]]
        return all(fn is not None and torch._C._will_engine_execute_node(fn) for fn in nodes)
    except Exception:      # "a leaf node was passed ... while running autograd.grad": captures, not accumulation
        return False


DEFAULT = object()       # "no per-call choice": the process-wide attach() state applies


def sh_sink_for(node_ctx, count, sh_tensors, explicit=None):
    """The attached ``dist_views.FactoredShGrad`` when the running backward pass may leave its SH gradient there: the
    SH inputs are leaves (``finish`` writes their ``.grad``; behind a torch ``cat`` the rows are needed here) and the
    engine accumulates into the node's ``count`` leaves (a ``.backward()`` of a training step -- never under
    ``torch.autograd.grad``, whose caller expects the rows returned)."""
    # explicit = (sink, exchange) of the call's RenderOptions; None: whatever is attached process-wide
    sink, hook = (_sh_sink, _exchange_hook) if explicit is None else explicit
    if sink is None or not all(t.is_leaf and t.requires_grad for t in sh_tensors) or \
            not _engine_accumulates_into(node_ctx, count):
        return None
    if hook is not None:
        raise RuntimeError("FactoredShGrad and ChunkedExchange cannot be attached together (the overlapped exchange "
                           "all-reduces the SH rows the factored form never writes)")
    return sink


def accumulation_targets(leaves, node_ctx=None, count=None, explicit=None):
    """The ``.grad`` tensors of ``leaves`` when the coming backward may add to them in place (see
    ``accumulate_in_kernel``), else None.  ``node_ctx``: the autograd node whose backward is running -- the in-kernel
    accumulation is only taken when the engine itself would accumulate into every leaf (never under
    ``torch.autograd.grad``, whose callers expect returned tensors and an untouched ``.grad``).  ``count``: how many
    inputs of the node are differentiated leaves (default ``len(leaves)``; larger when ``leaves`` leaves the SH
    tensors out because their gradient goes to a ``FactoredShGrad``)."""
    # explicit = (accumulate, exchange) of the call's RenderOptions; None: the process-wide block / attach() state
    on, hook = (getattr(_acc_flag, "on", False), _exchange_hook) if explicit is None else explicit
    if not on or hook is not None:
        return None
    if node_ctx is not None and not _engine_accumulates_into(node_ctx, len(leaves) if count is None else count):
        return None
    grads = []
    for t in leaves:
        g = t.grad if (t.is_leaf and t.requires_grad) else None
        if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != t.shape or \
                (g.data_ptr() & 15) or g.device != t.device:
            return None
        grads.append(g)
    if flat_grad_buffer(leaves) is None:      # not the one-buffer layout this module's backward hands out
        return None
    return grads


def backward(pws, shs, alphas, scales, rots, cam, S: FusedState, dloss_dgammas, high_shs=None, accumulate=None,
             sh_sink=None, exchange=DEFAULT):
    """-> (dloss_dpws[N,3], dloss_dshs[N,K], dloss_dalphas[N,1], dloss_dscales[N,3],
           dloss_drots[N,4], dloss_dus[N,2])  -- the gradient tuple of gsmodel.py:87-93.
    With ``high_shs`` (raw tensors, see ``forward``): -> (dpws, dlow_shs[N,3], dhigh_shs[N,K-3],
    dalphas_raw[N,1], dscales_raw, drots_raw, dus).
    ``accumulate``: the five (raw: six) gradient tensors of earlier views, in the order of the return tuple; this
    view's gradients are ADDED to them by the kernel and the same tensors are returned.
    ``sh_sink`` (``dist_views.FactoredShGrad``): the SH gradient of this view is left there as dL/dcolour [N,3]
    (``EGS_BWD_FACTORED_SH``); the SH entries of the return tuple are None, ``accumulate`` holds the other four
    tensors only, and the flat buffer is the 11 floats per Gaussian of pws, alphas, scales, rots."""
    raw = high_shs is not None
    pws = _chk(pws, "pws", torch.float32, (None, 3))
    n = pws.shape[0]
    if raw:
        shs, high_shs, K = _split_sh(shs, high_shs, n)
    else:
        shs = _chk(shs, "shs", torch.float32, (n, None))
        K = shs.shape[1]
    alphas = _alphas(alphas, n)
    scales = _chk(scales, "scales", torch.float32, (n, 3))
    rots = _chk(rots, "rots", torch.float32, (n, 4))
    W, H = S.width, S.height
    dl = _chk(dloss_dgammas, "dloss_dgammas", torch.float32, (3, H, W))
    lib = _lib_on(pws)
    dev = pws.device
    f32 = torch.float32
    # The parameter gradients are slices of ONE allocation (order: pws, shs | low, high, alphas, scales,
    # rots): a data-parallel caller exchanges all 59 floats per Gaussian with a single all-reduce of
    # ``flat_grad_buffer(params)`` instead of five or six latency-bound ones (autograd adopts the slices as
    # ``.grad`` without copying).
    widths = [3, 3, K - 3, 1, 3, 4] if raw else [3, K, 1, 3, 4]
    if sh_sink is not None:
        widths = [3, 1, 3, 4]
    if accumulate is not None:
        parts = [g.view(n, w) for g, w in zip(accumulate, widths)]
    else:
        starts, at = [], 0
        for w in widths:                      # every slice starts 16-B aligned (the kernels store dwordx4)
            starts.append(at)
            at += (n * w + 3) // 4 * 4
        flat = torch.empty(at, dtype=f32, device=dev)
        # the <= 3 alignment words behind a slice belong to nobody: zero, not whatever the allocator left there --
        # whole-buffer operations (ViewStreams.finish adds flat buffers, callers all-reduce / norm / isfinite them)
        # must never meet NaN garbage.  No padding (and no kernel) when N is a multiple of four.
        pad = [i for a, w in zip(starts, widths) for i in range(a + n * w, a + (n * w + 3) // 4 * 4)]
        if pad:
            pk = (dev, n, tuple(widths))
            idx = _pad_index.get(pk)
            if idx is None:
                # (one index tensor per device and layout: a densifying trainer changes N every few epochs, and the
                # entries of the sizes it left behind would pile up)
                for old in [q for q in _pad_index if q[0] == dev and q[2] == pk[2]]:
                    del _pad_index[old]
                idx = _pad_index[pk] = torch.tensor(pad, dtype=torch.int64, device=dev)
            flat.index_fill_(0, idx, 0.0)
        parts = [flat[a:a + n * w].view(n, w) for a, w in zip(starts, widths)]
    if sh_sink is not None:
        dpws, dalphas, dscales, drots = parts
        dshs, dhigh = sh_sink.slot(n, K, cam), None     # [N,3]: dL/dcolour of this view (written, never added to)
    elif raw:
        dpws, dshs, dhigh, dalphas, dscales, drots = parts
    else:
        dpws, dshs, dalphas, dscales, drots = parts
        dhigh = None
    dus = torch.empty((n, 2), dtype=f32, device=dev)
    ws_bytes = lib.egs_fused_backward_ws_bytes(n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    mid = (_ptr(alphas), _ptr(cam.Rcw), _ptr(cam.tcw), _ptr(cam.twc), float(cam.fx), float(cam.fy), float(cam.cx),
           float(cam.cy), C.byref(_pol()), _ptr(S.us), _ptr(S.cinv2ds), _ptr(S.colors), _ptr(S.areas), _ptr(S.rec),
           _ptr(S.depths), _ptr(S.contrib), _ptr(S.final_tau), _ptr(S.ranges), _ptr(S.gsid), _ptr(dl), _ptr(ws),
           ws_bytes, _ptr(dpws), _ptr(dshs))
    st = _stream()
    gpack, S.gpack = S.gpack, None      # zeroed by the forward draw kernel: good for ONE backward pass
    seg = getattr(S, "seg", None)       # the forward pass split its long lists: the backward pass walks its segments
    seg_bytes = seg.numel() if seg is not None else 0
    if raw:
        launch = lambda phase, b, c: _lib.check(lib.egs_fused_backward_raw(
            n, K, S.gsid.shape[0], W, H, _ptr(pws), _ptr(rots), _ptr(scales), _ptr(shs), _ptr(high_shs), *mid,
            _ptr(dhigh), _ptr(dalphas), _ptr(dscales), _ptr(drots), _ptr(dus), _ptr(S.order), _ptr(gpack),
            _ptr(getattr(S, "dcw", None)), phase, b, c, _ptr(seg), seg_bytes, st))
    else:
        launch = lambda phase, b, c: _lib.check(lib.egs_fused_backward(
            n, K, S.gsid.shape[0], W, H, _ptr(pws), _ptr(rots), _ptr(scales), _ptr(shs), *mid, _ptr(dalphas),
            _ptr(dscales), _ptr(drots), _ptr(dus), _ptr(S.order), _ptr(gpack), _ptr(getattr(S, "dcw", None)), phase,
            b, c, _ptr(seg), seg_bytes, st))
    # the forward pass was dispatched by remembered work: the backward pass keeps its order (no second order kernel)
    keep = KEEP_FORWARD_ORDER if (REUSE_ORDER and getattr(S, "order_by_work", False)) else 0
    if getattr(S, "culled", False):
        keep |= CULLED_LISTS          # the list values carry block masks
    if accumulate is not None:
        keep |= ACCUMULATE            # the outputs hold earlier views' gradients: add to them
    if sh_sink is not None:
        keep |= FACTORED_SH
    hook = _exchange_hook if exchange is DEFAULT else exchange      # (``exchange``: the call's own ChunkedExchange or None)
    if hook is not None and sh_sink is not None:
        raise RuntimeError("fused.backward: sh_sink and an attached ChunkedExchange exclude each other")
    chunks = hook.chunks if hook is not None else 1
    rows = -(-n // (256 * chunks)) * 256 if chunks > 1 else n     # rows per chunk: whole workgroups
    if hook is not None:
        hook.begin_backward()              # one backward pass per attach()/finish(): raises on a second one
    if hook is None or chunks <= 1 or rows >= n:
        launch(0 | keep, 0, n)
        if hook is not None:
            # FRESH view objects: a second reference to the tensors returned below would make AccumulateGrad
            # clone them, and the reduced values would never reach .grad
            hook.on_chunk([p[:] for p in parts])
    else:
        # The chain rule runs in a few row chunks; each chunk's gradient slices go to the exchange as soon as
        # its kernel is enqueued, so the all-reduce of chunk k overlaps the computation of chunk k + 1
        launch(1 | keep, 0, 0)
        for b in range(0, n, rows):
            c = min(rows, n - b)
            launch(2 | (keep & (ACCUMULATE | FACTORED_SH)), b, c)
            hook.on_chunk([p[b:b + c] for p in parts])
    if sh_sink is not None:
        return (dpws, None, None, dalphas, dscales, drots, dus) if raw else (dpws, None, dalphas, dscales, drots, dus)
    if raw:
        return dpws, dshs, dhigh, dalphas, dscales, drots, dus
    return dpws, dshs, dalphas, dscales, drots, dus
