"""Drop-in replacement of the reference's CUDA extension module ``gsplatcu``.

Same seven ops, same positional signatures, same return lists as the pybind11
module built from reference gsplatcu/ext.cpp:68-77 (host wrappers
gsplatcu/gausplat.cu), so the reference's callers -- forward_gpu.py:47-60,
backward_gpu.py:81-138, gsplat/gsmodel.py:21-39,67-69 -- run unmodified with

    import easygaussiansplatting_amd.gsplatcu as gsc      # or: import gsplatcu as gsc

Underneath, every op is one (or a few) hand-written HIP kernels for gfx950 in
``libegs_hip.so`` reached through the C ABI of include/egs_hip.h; this module is
only host plumbing: validation, output allocation from torch's caching
allocator, the current HIP stream.  Differences from the reference, all
supersets:

* work is enqueued on ``torch.cuda.current_stream()`` with no device-wide
  synchronisation (the reference syncs after every kernel, common.cuh:16-24);
  ``splat`` performs exactly one 8-byte device->host read (the patch count and the largest depth key),
  where the reference has one too (gausplat.cu:67);
* bad dtype / device / shape raise ``ValueError``/``TypeError`` instead of
  reading out of bounds; HIP errors raise ``RuntimeError``;
* ``N == 0`` and ``P == 0`` are legal (the reference crashes at gausplat.cu:67);
* the raster policy is module state: ``set_policy("gsplatcu")``
  (default, the CUDA extension's semantics) or ``set_policy("forward_cpu")``.
"""
from __future__ import annotations

import ctypes as C
import threading

import torch

from . import _lib
from ._lib import EgsPolicy

__all__ = ["project", "computeCov3D", "computeCov2D", "sh2Color", "inverseCov2D", "splat", "splatB",
           "set_policy", "get_policy", "chain_rule", "clear_memo", "set_memo", "splat_with_records", "SplatRecords"]

_policy_name = "gsplatcu"
_policy = None


def _pol() -> EgsPolicy:
    global _policy
    if _policy is None:
        set_policy(_policy_name)
    return _policy


def set_policy(name: str) -> None:
    """Select which of the reference's pipeline definitions the ops follow:
    ``"gsplatcu"`` (gsplatcu/kernel.cu; default) or ``"forward_cpu"``
    (gsplat/gausplat.py as driven by forward_cpu.py).  ``"gsplatcu_nan_skip"``: the default with one opt-in
    deviation -- a Gaussian whose conic holds a NaN is skipped instead of blended at min(0.99, alpha) (the CUDA
    extension's ``max(0.0f, NaN) == 0``, kernel.cu:243-246): no NaN colour can reach the image."""
    global _policy, _policy_name
    lib = _lib.load()
    p = EgsPolicy()
    if name in ("gsplatcu", "gsplatcu_nan_skip"):
        lib.egs_policy_gsplatcu(C.byref(p))
        p.nan_maha = 1 if name == "gsplatcu_nan_skip" else 0
    elif name == "forward_cpu":
        lib.egs_policy_forward_cpu(C.byref(p))
    else:
        raise ValueError("unknown raster policy %r (expected 'gsplatcu', 'gsplatcu_nan_skip' or 'forward_cpu')" % (name,))
    _policy, _policy_name = p, name


def get_policy() -> str:
    return _policy_name


# ------------------------------------------------------------------ helpers
def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk(t, name, dtype, shape):
    """dtype/device/shape validation; returns a contiguous tensor (a copy only
    if the caller's tensor was not contiguous, like the reference's .contiguous())."""
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor, got %s" % (name, type(t).__name__))
    if not t.is_cuda:
        raise ValueError("%s must live on the GPU (got device %s)" % (name, t.device))
    if t.dtype != dtype:
        raise ValueError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if len(shape) != t.dim() or any(s is not None and s != d for s, d in zip(shape, t.shape)):
        raise ValueError("%s must have shape %s, got %s" % (name, list(shape), list(t.shape)))
    return t.contiguous()


def _out(shape, like, dtype=torch.float32):
    """Output of an op: the kernels write EVERY row (culled Gaussians as zeros, what the reference's zero-filled
    ``torch::full(..., 0)`` outputs read as, gausplat.cu:170-178), so no fill kernel runs -- torch.zeros here cost
    528 B per Gaussian and training step of pure memset."""
    return torch.empty(shape, dtype=dtype, device=like.device)


def _lib_on(t):
    lib = _lib.load()
    if t.device.index is not None and t.device.index != torch.cuda.current_device():
        raise ValueError("tensors live on %s but the current device is cuda:%d"
                         % (t.device, torch.cuda.current_device()))
    return lib


# ------------------------------------------------------------------ the seven ops
def project(pws, Rcw, tcw, focal_x, focal_y, center_x, center_y, calc_J):
    """-> [us[N,2], pcs[N,3], depths[N]] (+ [du_dpcs[N,2,3]] if calc_J).
    Reference: ext.cpp:54-61, gausplat.cu:253-296, kernel.cu:553-617."""
    pws = _chk(pws, "pws", torch.float32, (None, 3))
    Rcw = _chk(Rcw, "Rcw", torch.float32, (3, 3))
    tcw = _chk(tcw, "tcw", torch.float32, (3,))
    lib = _lib_on(pws)
    n = pws.shape[0]
    us = _out((n, 2), pws); pcs = _out((n, 3), pws); depths = _out((n,), pws)
    J = _out((n, 2, 3), pws) if calc_J else None
    _lib.check(lib.egs_project(n, _ptr(pws), _ptr(Rcw), _ptr(tcw), float(focal_x), float(focal_y),
                               float(center_x), float(center_y), C.byref(_pol()), _ptr(us), _ptr(pcs),
                               _ptr(depths), _ptr(J), _stream()))
    return [us, pcs, depths, J] if calc_J else [us, pcs, depths]


def computeCov3D(rots, scales, depths, calc_J):
    """-> [cov3ds[N,6]] (+ [dcov3d_drots[N,6,4], dcov3d_dscales[N,6,3]]).
    Reference: ext.cpp:39-42, gausplat.cu:162-199, kernel.cu:326-423."""
    rots = _chk(rots, "rots", torch.float32, (None, 4))
    n = rots.shape[0]
    scales = _chk(scales, "scales", torch.float32, (n, 3))
    depths = _chk(depths, "depths", torch.float32, (n,))
    lib = _lib_on(rots)
    cov3ds = _out((n, 6), rots)
    dq = _out((n, 6, 4), rots) if calc_J else None
    ds = _out((n, 6, 3), rots) if calc_J else None
    _lib.check(lib.egs_cov3d(n, _ptr(rots), _ptr(scales), _ptr(depths), C.byref(_pol()), _ptr(cov3ds),
                             _ptr(dq), _ptr(ds), _stream()))
    return [cov3ds, dq, ds] if calc_J else [cov3ds]


def computeCov2D(cov3ds, pcs, Rcw, depths, focal_x, focal_y, width, height, calc_J):
    """-> [cov2ds[N,3]] (+ [dcov2d_dcov3ds[N,3,6], dcov2d_dpcs[N,3,3]]).
    NOTE the argument order ``width, height`` (ext.cpp:44-52); the reference's
    forward_gpu.py:53 passes them swapped.  Reference: gausplat.cu:201-251,
    kernel.cu:425-551."""
    pcs = _chk(pcs, "pcs", torch.float32, (None, 3))
    n = pcs.shape[0]
    cov3ds = _chk(cov3ds, "cov3ds", torch.float32, (n, 6))
    Rcw = _chk(Rcw, "Rcw", torch.float32, (3, 3))
    depths = _chk(depths, "depths", torch.float32, (n,))
    lib = _lib_on(pcs)
    cov2ds = _out((n, 3), pcs)
    d3 = _out((n, 3, 6), pcs) if calc_J else None
    dpc = _out((n, 3, 3), pcs) if calc_J else None
    _lib.check(lib.egs_cov2d(n, _ptr(cov3ds), _ptr(pcs), _ptr(Rcw), _ptr(depths), float(focal_x),
                             float(focal_y), float(width), float(height), C.byref(_pol()), _ptr(cov2ds),
                             _ptr(d3), _ptr(dpc), _stream()))
    return [cov2ds, d3, dpc] if calc_J else [cov2ds]


def sh2Color(shs, pws, twc, calc_J):
    """-> [colors[N,3]] (+ [dcolor_dshs[N,1,K/3], dcolor_dpws[N,3,3]]).
    shs[N,K], K in {3,12,27,48}, layout sh[i, 3*c + rgb].
    Reference: ext.cpp:63-66, gausplat.cu:298-338, kernel.cu:619-807."""
    pws = _chk(pws, "pws", torch.float32, (None, 3))
    n = pws.shape[0]
    shs = _chk(shs, "shs", torch.float32, (n, None))
    K = shs.shape[1]
    if K not in (3, 12, 27, 48):
        raise ValueError("shs must have 3, 12, 27 or 48 columns (SH degree 0..3), got %d" % K)
    twc = _chk(twc, "twc", torch.float32, (3,))
    lib = _lib_on(pws)
    colors = _out((n, 3), pws)
    dsh = _out((n, 1, K // 3), pws) if calc_J else None
    dpw = _out((n, 3, 3), pws) if calc_J else None
    _lib.check(lib.egs_sh2color(n, K, _ptr(shs), _ptr(pws), _ptr(twc), _ptr(colors), _ptr(dsh), _ptr(dpw),
                                _stream()))
    return [colors, dsh, dpw] if calc_J else [colors]


def inverseCov2D(cov2ds, depths, calc_J):
    """-> [cinv2ds[N,3], areas[N,2] int32] (+ [dcinv2d_dcov2ds[N,3,3]]).
    ``depths`` is updated IN PLACE (NaN determinant -> -1) as in the reference
    (kernel.cu:300-305).  Reference: ext.cpp:34-36, gausplat.cu:340-373."""
    cov2ds = _chk(cov2ds, "cov2ds", torch.float32, (None, 3))
    n = cov2ds.shape[0]
    depths = _chk(depths, "depths", torch.float32, (n,))
    lib = _lib_on(cov2ds)
    cinv = _out((n, 3), cov2ds)
    areas = _out((n, 2), cov2ds, torch.int32)
    J = _out((n, 3, 3), cov2ds) if calc_J else None
    _lib.check(lib.egs_inv_cov2d(n, _ptr(cov2ds), _ptr(depths), C.byref(_pol()), _ptr(cinv), _ptr(areas),
                                 _ptr(J), _stream()))
    return [cinv, areas, J] if calc_J else [cinv, areas]


_key_bits = {}   # (device index, problem key) -> significant depth-key bits learnt from the previous call


def _get_key_bits(dev_index, key=None) -> int:
    return _key_bits.get((dev_index, key), 32)


def _set_key_bits(dev_index, key, bits) -> None:
    _key_bits[(dev_index, key)] = int(bits)
    _key_low.pop((dev_index, key), None)


_key_low = {}    # (device index, problem key) -> [renders in a row that needed fewer bits, the most they needed]
KEY_BITS_DECAY = 32   # renders in a row with a smaller need before the hint is lowered


def _learn_key_bits(dev_index, key, need, missed=False) -> None:
    """Update the depth-key bit hint of a problem size from one render's largest key (``need`` bits).
    The hint is shared by all cameras that render this problem size, so it follows a slowly decaying MAXIMUM:
    raised at once, lowered only after KEY_BITS_DECAY renders in a row needed less (to the most they needed) --
    cameras whose depth ranges differ by a bit or two then never miss, where "need + 1 after every success"
    made them alternate between a miss (a redone step under deferred validation) and a reset.  A miss sets
    the hint to 32 for the redo; the first success after that adopts need + 1."""
    k = (dev_index, key)
    target = min(32, int(need) + 1)
    while len(_key_bits) > 1024:          # (bounded like the capacity table: fused.SIZE_TABLE_MAX)
        old = next(iter(_key_bits))
        _key_bits.pop(old, None)
        _key_low.pop(old, None)
    if missed:
        _key_bits[k] = 32
        _key_low.pop(k, None)
        return
    hint = _key_bits.get(k, 32)
    if hint >= 32 or target >= hint:
        _key_bits[k] = target
        _key_low.pop(k, None)
        return
    low = _key_low.setdefault(k, [0, 0])
    low[0] += 1
    low[1] = max(low[1], target)
    if low[0] >= KEY_BITS_DECAY:
        _key_bits[k] = low[1]
        _key_low.pop(k, None)


def _bin_stage(enqueue, device, key=None, while_waiting=None):
    """Run the binning stage with the depth-key bit-count hint protocol of egs_splat_bin:
    ``enqueue(hint, total)`` enqueues the stage; returns the patch count P.  The single
    8-byte read-back (reference: gausplat.cu:67) also brings the largest depth key, which
    sizes the next call's sort (depth keys rarely need more than 16 of their 32 bits; the hint is kept
    per device and problem ``key``); a too-small hint triggers one full-width re-run.
    ``while_waiting()`` runs between the enqueue and the blocking read: host work that does not need
    P (output allocations) belongs there, so that the GPU idles as briefly as possible afterwards."""
    total = torch.empty(2, dtype=torch.int32, device=device)
    hint = _get_key_bits(device.index, key)
    enqueue(hint, total)
    if while_waiting is not None:
        while_waiting()
    p, mk = (int(v) & 0xFFFFFFFF for v in total.tolist())
    need = mk.bit_length()
    if hint < 32 and need > hint:
        enqueue(32, total)
        p, mk = (int(v) & 0xFFFFFFFF for v in total.tolist())
    _learn_key_bits(device.index, key, need)
    if p >= 2**31:
        raise RuntimeError("splat: %d tile patches overflow int32 indexing" % p)
    return p


def _tiles(width, height):
    return ((width + 15) // 16) * ((height + 15) // 16)


def _alphas(alphas, n):
    if not isinstance(alphas, torch.Tensor) or alphas.numel() != n:
        raise ValueError("alphas must be a tensor of shape [N] or [N,1] with N=%d" % n)
    return _chk(alphas.reshape(n), "alphas", torch.float32, (n,))


# ---- what `splat` keeps for the `splatB` that follows it ------------------------------------------------------
# GSFunction.backward (gsmodel.py:67-69) calls splatB with the very tensors its forward gave to splat: the list with
# block masks the forward draw walked, the draw's [dispatch order | measured work] buffer and an [N][12] gradient-record
# buffer the forward draw kernel cleared on the side can be reused instead of rebuilt.  Two forms:
#
#   * the PUBLIC pair (what an unmodified reference GSFunction calls), CONTENT-VALIDATED (round 4, second half):
#     `splat` keeps, per (device, stream), the masked list, the order buffer, the cleared gradient records and a STAMP
#     of the us / cinv2ds / alphas VALUES (two position-dependent 32-bit sums per 256 Gaussians, written by the kernel
#     that packs them).  `splatB` always packs its records from the tensors it is given, stamps them too, and a kernel
#     repairs the kept list on the device: an entry that is not the caller's own (gsid_per_patch[i] differs) or whose
#     Gaussian lies in a block with a different stamp becomes the caller's entry with all four blocks set.  No pointer
#     or version is compared: a write through ``tensor.data``, another library's kernel, other tensors with the same
#     values -- all handled by what the VALUES are.  The order buffer (any permutation is correct) and the cleared
#     gradient records (data-independent, handed out once) need no validation.  OPT-IN since round 5
#     (``set_memo(True)``): it is worth 1 % of a seven-op step (1.147-1.150 against 1.159 ms) and costs ~80 MB per
#     (device, stream) at 1 M Gaussians of module-global state that a drop-in's callers know nothing about; at most
#     ``MEMO_MAX`` (device, stream) entries are kept, the least recently used goes first.
#   * the explicit HANDLE for a caller that OWNS the tensors between the two calls: ``splat_with_records`` returns a
#     ``SplatRecords`` and ``splatB(..., records=handle)`` takes it back -- this package's GSFunction (mode "ops") does
#     that with its own intermediates (us / cinv2ds / colors never leave the autograd node) and skips both the re-pack
#     and the validation (21 + 10 us); the handle is checked by (data_ptr, _version, shape), policy, size and stream.
#   * round 6, long lists only: when the public ``splat`` split its long tile lists over waves it keeps that
#     draw's SEGMENT-END STATES (G_s, T_end per segment and pixel: ~6 KB per 256 entries of a split tile), one entry per
#     (device, stream), replaced by the next ``splat`` there -- without them ``splatB`` rebuilds them from ``contrib``
#     first, a forward draw's worth of work (2.69 ms per step on scene.skewed_scene(reset_alpha=True)).  Three settings
#     (``set_pair_states``):
#       "content" (default)  the states AND one snapshot buffer of the eight tensors the pair shares (us, cinv2ds, alphas,
#                 colors in; contrib, final_tau, patch_range_per_tile, gsid_per_patch out: 4 (9 N + 2 HW + 2 T + P) bytes,
#                 112 MB on that scene).  ``splatB`` enqueues a bitwise comparison of what it was HANDED with the snapshot
#                 (egs_words_differ: one pass), then the backward pass of the SNAPSHOT from the kept states -- self-
#                 consistent whatever happened to the caller's tensors --, and only then looks at the verdict (one
#                 page-locked word; the GPU is busy with the backward kernels meanwhile).  Equal: done, 2.31 ms.  Any value
#                 differs (a write through ``tensor.data``, which no version counter sees; other tensors): everything again
#                 from the handed tensors, as with nothing kept.  The pair stays a pure function of its arguments; clones
#                 holding the same values find the states too.
#       True      states only, matched by (data_ptr, in-place version) of the eight tensors, the contract of the records
#                 handle: nothing is compared (2.21 ms), a write through ``.data`` between the two calls goes unseen --
#                 the forced-segments sweep of the suite found exactly that in tests/test_gpu_memo.py's ``data_write``.
#       False     nothing kept.
#     Scenes whose walks stay below the split threshold never create an entry.
_memo_enabled = False
_splat_memo = {}        # (device, stream) -> SplatRecords, in order of last use
MEMO_MAX = 8
_pair_states_enabled = "content"     # "content" (default) | True (identity + version, unvalidated) | False (nothing kept)
_pair_states = {}       # (device, stream) -> dict(outs, sig, in_sig, seg, lists, width, height, policy)
_last_splatB = {"segments": False, "rebuilt": False, "kept_states": False}
_pair_tls = threading.local()   # hands the segment workspace of a _splat call to the public splat() around it
MASKED_LISTS = True     # A/B knob: exact block masks in the seven-op surface's list values (egs_splat_bin_pack)


class SplatRecords:
    """Opaque: what one ``splat`` call left for the ``splatB`` of the same tensors (see above)."""
    __slots__ = ("tensors", "sig", "width", "height", "policy", "rec", "order", "gpack", "dev_index", "stream",
                 "lists", "pair", "pair_sig", "stamp", "n", "npatch", "visible", "seg")

    def matches(self, dev, st, tensors, width, height):
        if (self.dev_index != dev.index or self.stream != int(st.value or 0) or self.width != width
                or self.height != height or self.policy != _policy_name):
            return False
        sig = _memo_sig(tensors)
        return sig is not None and sig == self.sig and _memo_sig(self.tensors) == self.sig


def set_memo(on: bool) -> None:
    """Switch the content-validated keeping of the public ``splat`` -> ``splatB`` pair on or off (default: off)."""
    global _memo_enabled
    _memo_enabled = bool(on)
    if not on:
        _splat_memo.clear()


def set_pair_states(on):
    """What the public ``splat`` keeps of a draw that split its long tile lists (only such scenes ever hold anything) for
    the ``splatB`` that follows.  ``"content"`` (default): the segment-end states AND a snapshot of the eight tensors the
    pair shares (four inputs, four outputs; 4 (9 N + 2 HW + 2 T + P) bytes); ``splatB`` differentiates that snapshot from
    the kept states at once, compares what it was HANDED with the snapshot on the device meanwhile, and only if a value
    differs (a write through ``.data``, other tensors) does everything again from the tensors it was handed, as by
    default before: a pure function of its arguments either way.  ``True``: states only, matched by data_ptr and in-place
    version (nothing compared: for callers that never write through ``tensor.data`` between the two calls; 0.1 ms less).
    ``False``: nothing kept, ``splatB`` always rebuilds the states from ``contrib``.  -> the previous setting."""
    global _pair_states_enabled
    prev = _pair_states_enabled
    _pair_states_enabled = "content" if on == "content" else bool(on)
    _pair_states.clear()
    return prev


_pair_flags = {}        # (device, stream) -> page-locked bool[1] the verdict of a content comparison lands in


def _pair_flag(key):
    f = _pair_flags.get(key)
    if f is None:
        while len(_pair_flags) >= 4 * MEMO_MAX:
            _pair_flags.pop(next(iter(_pair_flags)))
        f = _pair_flags[key] = torch.empty(1, dtype=torch.int32).pin_memory()
    return f


def _snapshot(tensors):
    """One int32 buffer holding the bits of ``tensors`` (4-byte dtypes, contiguous), every piece 16-B aligned (the
    comparison then reads sixteen bytes per lane), + the views of it in their shapes."""
    flat, pad = [], torch.zeros(3, dtype=torch.int32, device=tensors[0].device)
    for t in tensors:
        f = t.reshape(-1).view(torch.int32)
        flat.append(f)
        if f.numel() % 4:
            flat.append(pad[:4 - f.numel() % 4])
    buf = torch.cat(flat)
    views, at = [], 0
    for t in tensors:
        views.append(buf[at:at + t.numel()].view(t.dtype).reshape(t.shape))
        at += (t.numel() + 3) // 4 * 4
    return buf, views


def last_splatB_info() -> dict:
    """What the last ``splatB`` of this process did: ``segments`` (long lists walked by one wave per segment), ``rebuilt``
    (the segment states were rebuilt from ``contrib``), ``kept_states`` (the forward's states were found).  For tests
    and bench lines."""
    return dict(_last_splatB)


def clear_memo() -> None:
    """Drop what ``splat`` keeps for the next ``splatB`` (per device and stream: the list with masks, the dispatch-order
    buffer, the cleared gradient records and the content stamps -- ~80 MB at 1 M Gaussians until the next ``splat``)."""
    _splat_memo.clear()
    _pair_states.clear()


def _memo_sig(tensors):
    """(data_ptr, in-place version, shape) per tensor, or None where torch keeps no version counter (tensors created
    under ``torch.inference_mode()``): no signature, nothing is kept, ``splat`` itself works as always."""
    try:
        return tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in tensors)
    except RuntimeError:
        return None


def _make_records(dev, st, tensors, width, height, rec, order, gpack, lists=None, pair=None, stamp=None, n=0,
                  npatch=-1):
    sig = _memo_sig(tensors) if tensors is not None else ()
    if sig is None:
        return None
    h = SplatRecords()
    h.stamp, h.n, h.npatch, h.visible, h.seg = stamp, n, npatch, None, None
    h.tensors, h.sig, h.width, h.height, h.policy = tensors, sig, width, height, _policy_name
    h.rec, h.order, h.gpack, h.dev_index, h.stream = rec, order, gpack, dev.index, int(st.value or 0)
    # the list WITH block masks the forward draw walked, valid for the (gsid_per_patch, patch_range_per_tile) pair
    # this splat returned -- and only for it
    h.lists, h.pair, h.pair_sig = lists, pair, (_memo_sig(pair) if pair is not None else None)
    if tensors is not None and h.pair_sig is None:   # (handle form: the masked list is tied to its returned pair)
        h.lists = None
    return h


def _walked_lists(h, gsid, ranges):
    """The masked list of the handle if ``gsid`` / ``ranges`` are the pair its splat returned (same memory, same
    version), else None: splatB then walks the caller's plain list with the per-entry box test."""
    if h is None or h.lists is None:
        return None
    sig = _memo_sig((gsid, ranges))
    return h.lists if (sig is not None and sig == h.pair_sig and _memo_sig(h.pair) == sig) else None


def _take_records(h, dev, st, tensors, width, height):
    """-> (rec, order, gpack) of a handle that still describes ``tensors``; gpack (cleared by the forward draw) is
    handed out ONCE."""
    if h is None or not h.matches(dev, st, tensors, width, height):
        return None, None, None
    gpack, h.gpack = h.gpack, None
    return h.rec, h.order, gpack


def splat_with_records(height, width, us, cinv2ds, alphas, depths, colors, areas):
    """``splat`` + a ``SplatRecords`` handle (or None) for ``splatB(..., records=handle)``.  For callers that own
    the four input tensors until that ``splatB`` (see the note above ``SplatRecords``)."""
    return _splat(height, width, us, cinv2ds, alphas, depths, colors, areas, keep="handle")


def splat(height, width, us, cinv2ds, alphas, depths, colors, areas):
    """-> [image[3,H,W], contrib[H,W] int32, final_tau[H,W],
           patch_range_per_tile[T,2] int32, gsid_per_patch[P] int32].
    ``depths`` and ``areas`` are updated IN PLACE for Gaussians whose tile rect is
    empty (kernel.cu:114-119).  Reference: ext.cpp:10-18, gausplat.cu:24-112."""
    out, h = _splat(height, width, us, cinv2ds, alphas, depths, colors, areas, keep="public" if _memo_enabled else False)
    if _pair_states_enabled:
        key = (us.device.index, int(_stream().value or 0))
        _pair_states.pop(key, None)
        kept = getattr(_pair_tls, "seg_of_last_call", None)
        _pair_tls.seg_of_last_call = None
        if kept is not None:
            # (the inputs as splatB will see them: it normalises ``alphas`` [N,1] -> [N] before it forms its signature --
            # through GSFunction, which hands both calls the [N,1] tensor, nothing ever matched before this line said so)
            try:
                in_sig = _memo_sig((us, cinv2ds, _alphas(alphas, us.shape[0]), colors))
            except Exception:
                in_sig = None
            sig = _memo_sig(out[1:5])
            ent = None
            if _pair_states_enabled == "content":
                try:     # the eight tensors as splatB will be handed them (alphas [N]; all 4-byte types, contiguous)
                    n_ = us.shape[0]
                    eight = (_chk(us, "us", torch.float32, (n_, 2)), _chk(cinv2ds, "cinv2ds", torch.float32, (n_, 3)),
                             _alphas(alphas, n_), _chk(colors, "colors", torch.float32, (n_, 3)),
                             out[1], out[2], out[3], out[4])
                    buf, views = _snapshot(eight)
                    ent = dict(snap=buf, views=views, flag=_pair_flag(key))
                except Exception:
                    ent = None
            elif sig is not None and in_sig is not None:
                ent = dict(outs=tuple(out[1:5]), sig=sig, in_sig=in_sig)
            if ent is not None:
                ent.update(seg=kept[0], lists=kept[1], width=int(width), height=int(height), policy=_policy_name)
                _pair_states[key] = ent
                while len(_pair_states) > MEMO_MAX:
                    _pair_states.pop(next(iter(_pair_states)))
    if _memo_enabled:
        key = (us.device.index, int(_stream().value or 0))
        _splat_memo.pop(key, None)          # (an older entry must not outlive its splat; re-inserted = most recent)
        if h is not None:
            _splat_memo[key] = h
            while len(_splat_memo) > MEMO_MAX:      # streams that are gone, one-off renders: least recently used first
                _splat_memo.pop(next(iter(_splat_memo)))
    return out


def _splat(height, width, us, cinv2ds, alphas, depths, colors, areas, keep):
    """-> ([image, contrib, final_tau, patch_range_per_tile, gsid_per_patch], SplatRecords or None)."""
    height, width = int(height), int(width)
    if height <= 0 or width <= 0:
        raise ValueError("height and width must be positive")
    us = _chk(us, "us", torch.float32, (None, 2))
    n = us.shape[0]
    cinv2ds = _chk(cinv2ds, "cinv2ds", torch.float32, (n, 3))
    alphas = _alphas(alphas, n)
    depths = _chk(depths, "depths", torch.float32, (n,))
    colors = _chk(colors, "colors", torch.float32, (n, 3))
    areas = _chk(areas, "areas", torch.int32, (n, 2))
    lib = _lib_on(us)
    dev = us.device
    pol = C.byref(_pol())
    image = torch.empty((3, height, width), dtype=torch.float32, device=dev)      # fully written by the callee
    contrib = torch.empty((height, width), dtype=torch.int32, device=dev)
    final_tau = torch.empty((height, width), dtype=torch.float32, device=dev)
    ranges = torch.empty((_tiles(width, height), 2), dtype=torch.int32, device=dev)
    ws_bin_bytes = lib.egs_splat_bin_ws_bytes(n)
    ws_bin = torch.empty(ws_bin_bytes, dtype=torch.uint8, device=dev)
    st = _stream()
    key = (n, width, height)
    # The packed 48-B records the draw kernels gather are built ONCE here; with keep=True they stay, with the
    # [dispatch order | measured work] buffer of the draw, for the splatB call of the same tensors (SplatRecords).
    rec = torch.empty((max(n, 1), 12), dtype=torch.float32, device=dev)
    order = torch.empty(lib.egs_tile_order_len(width, height), dtype=torch.int32, device=dev)
    gpack = None
    if not (n > 0 and _pol().footprint != 1):   # (pixel-box records also depend on `areas`, which this op mutates)
        keep = False
    # Tile-footprint policies with a skip threshold: records and binning state in ONE pass (egs_splat_bin_pack); the
    # lists are the reference's (gsid_per_patch bit-exact), their values carry exact 8x8-block masks the draw kernels
    # take instead of testing a box per entry.  The masked list stays internal; the caller gets the stripped copy.
    masks = MASKED_LISTS and n > 0 and _pol().footprint == 0 and _pol().alpha_skip > 0 and n < (1 << 28)
    flags = 2 if masks else 0            # EGS_DRAW_MASKED_LISTS
    if n > 0:
        if not masks:
            _lib.check(lib.egs_pack_records(n, width, height, _ptr(us), _ptr(cinv2ds), _ptr(alphas), _ptr(colors),
                                            _ptr(areas), pol, _ptr(rec), st))
        if keep:
            # the packed gradient records of a splatB that may follow: cleared on the side by the draw kernel (it is
            # VALU-bound, the memory system idles), good for ONE backward pass
            gpack = torch.empty((n, 12), dtype=torch.float32, device=dev)
    # Long tile lists split over several waves (include/egs_hip.h egs_splat_draw_rec_seg): taken when the longest walk a
    # recent call of this problem size reported exceeds the split threshold (page-locked hint words, read without
    # waiting; fused._seg_decision).  The op has no camera identity, so there is no walk on record per view: every
    # segment of a list is speculated when the scene's renders walk most of their lists, else segment 0 + one wave
    # continuing -- the BACKWARD pass is split either way (with the handle: from these states; public splatB: rebuilt).
    from . import fused as _fused            # (the per-device mailbox / capacity / hint state lives there)
    ctx = _fused._ctx(dev)
    use_seg, seg_hint = _fused._seg_decision(ctx, lib, key, _pol()) if n > 0 else (False, None)
    walk_word = _fused._walk_word(ctx, key, dev, st, seg_hint is not None)
    seg_flags = 8 if (use_seg and getattr(_fused._tls, "seg_speculate", False)) else 0     # EGS_DRAW_SEG_SPECULATE
    seg_ws = [None]
    lists = [None]     # the list the draw kernels walked (with masks), kept for the backward draw
    # content stamps of us / cinv2ds / alphas (public pair: splatB validates what it is given against them)
    stamp = torch.empty(lib.egs_pair_stamp_words(n), dtype=torch.int32, device=dev) if (masks and keep == "public") else None
    # depths > 0.2 after this op's in-place cull (the mask of gsmodel.py:50), written by the packing kernel on the side
    # for a caller that asked for the handle (this package's GSFunction): no separate compare kernel
    visible = torch.empty(n, dtype=torch.bool, device=dev) if (masks and keep == "handle") else None

    def records(gsid):    # only once the draw stage is enqueued: the order buffer is written, the gradient records cleared
        # (the public splat may keep this draw's segment states for the splatB of the tensors it returns: see above)
        _pair_tls.seg_of_last_call = (seg_ws[0], lists[0]) if (seg_ws[0] is not None and keep != "handle") else None
        if keep == "handle":
            h = _make_records(dev, st, (us, cinv2ds, alphas, colors), width, height, rec, order, gpack,
                              lists[0], (gsid, ranges) if lists[0] is not None else None)
            if h is not None:
                h.visible = visible
                h.seg = seg_ws[0]       # the segment states of this draw (long lists split over waves), if any
            return h
        if keep == "public" and lists[0] is not None:   # no tensor is referenced, no record kept: values are validated
            return _make_records(dev, st, None, width, height, None, order, gpack, lists[0], None, stamp, n,
                                 int(gsid.shape[0]))
        return None

    def enqueue_bin(hint, total, host_slot=None):
        if masks:
            _lib.check(lib.egs_splat_bin_pack(n, width, height, _ptr(us), _ptr(cinv2ds), _ptr(alphas), _ptr(colors),
                                              _ptr(areas), _ptr(depths), pol, hint, _ptr(ws_bin), ws_bin_bytes,
                                              _ptr(total), host_slot, _ptr(rec), _ptr(stamp), _ptr(visible), st))
        elif host_slot is not None:
            _lib.check(lib.egs_splat_bin_mb(n, width, height, _ptr(us), _ptr(areas), _ptr(depths), pol, hint,
                                            _ptr(ws_bin), ws_bin_bytes, _ptr(total), host_slot, st))
        else:
            _lib.check(lib.egs_splat_bin(n, width, height, _ptr(us), _ptr(areas), _ptr(depths), pol, hint,
                                         _ptr(ws_bin), ws_bin_bytes, _ptr(total), st))

    def draw_exact(patches, redo=False):
        # (``redo``: see fused.forward -- the second range kernel of a render clears the walk word without publishing it)
        gsid = torch.empty(patches, dtype=torch.int32, device=dev)
        walked = torch.empty(patches, dtype=torch.int32, device=dev) if masks else gsid
        ws_draw_bytes = lib.egs_splat_draw_ws_bytes(n, patches, width, height)
        ws_draw = torch.empty(ws_draw_bytes, dtype=torch.uint8, device=dev)
        # (with masks the range kernel also writes the plain list the caller gets: no strip launch)
        seg_ws[0] = torch.empty(lib.egs_seg_ws_bytes(max(patches, 1), width, height), dtype=torch.uint8,
                                device=dev) if use_seg else None
        _lib.check(lib.egs_splat_draw_rec_seg(n, patches, None, width, height, _ptr(rec), pol, _ptr(ws_bin),
                                              _ptr(ws_draw), ws_draw_bytes, _ptr(image), _ptr(contrib), _ptr(final_tau),
                                              _ptr(ranges), _ptr(walked), _ptr(order), _ptr(gpack), None, 0,
                                              flags | seg_flags, _ptr(seg_ws[0]),
                                              seg_ws[0].numel() if use_seg else 0,
                                              None if (redo and walk_word is not None) else seg_hint, _ptr(walk_word),
                                              _ptr(gsid) if masks else None, st))
        if masks:
            lists[0] = walked
        return gsid

    def render_exact(redo=False):
        """The reference's sequence (gausplat.cu:50-105): bin, read P back, draw -- the GPU idles around the read."""
        patches = _bin_stage(enqueue_bin, dev, key)
        return patches, draw_exact(patches, redo)

    # From the second call of a problem size on, the draw stage is enqueued BEHIND the binning stage before the
    # host has seen P: buffers sized by the largest count met so far (+6 %), the count taken from device memory,
    # {P, max depth key} delivered into a page-locked mailbox slot by the binning kernels.  `gsid_per_patch` must
    # come back with exactly P rows, so the host still waits for P -- while the GPU draws.
    cap = ctx.capacity.get(key, 0) if (_fused.ENQUEUE_AHEAD and n > 0) else 0
    slot = None
    if cap > 0:
        with ctx.lock:
            if ctx.free:
                slot = ctx.free.pop()
    if slot is None:
        patches, gsid = render_exact()
        if n > 0:
            with ctx.lock:
                _fused._learn_capacity(ctx, key, patches)
        return [image, contrib, final_tau, ranges, gsid], records(gsid)
    t = _fused._Ticket()
    t.ctx, t.key, t.cap, t.state, t.status, t.collected, t.slot = ctx, key, cap, None, _fused._Ticket.PENDING, True, slot
    t.hint = _get_key_bits(dev.index, key)
    try:
        total = torch.empty(2, dtype=torch.int32, device=dev)
        _lib.check(lib.egs_mailbox_arm(ctx.mb, slot, st))
        enqueue_bin(t.hint, total, C.c_void_p(lib.egs_mailbox_slot(ctx.mb, slot)))
        gsid_full = torch.empty(cap, dtype=torch.int32, device=dev)
        walked_full = torch.empty(cap, dtype=torch.int32, device=dev) if masks else gsid_full
        ws_draw_bytes = lib.egs_splat_draw_ws_bytes(n, cap, width, height)
        ws_draw = torch.empty(ws_draw_bytes, dtype=torch.uint8, device=dev)
        seg_ws[0] = torch.empty(lib.egs_seg_ws_bytes(cap, width, height), dtype=torch.uint8, device=dev) if use_seg \
            else None
        _lib.check(lib.egs_splat_draw_rec_seg(n, cap, _ptr(total), width, height, _ptr(rec), pol, _ptr(ws_bin),
                                              _ptr(ws_draw), ws_draw_bytes, _ptr(image), _ptr(contrib),
                                              _ptr(final_tau), _ptr(ranges), _ptr(walked_full), _ptr(order),
                                              _ptr(gpack), None, 0, flags | seg_flags, _ptr(seg_ws[0]),
                                              seg_ws[0].numel() if use_seg else 0, seg_hint, _ptr(walk_word),
                                              _ptr(gsid_full) if masks else None, st))
    except BaseException:
        # Kernels enqueued before the failure (the arm, the binning chain) still store {P, max key} into the slot:
        # it may only go back on the free list once they have run, or a later render that picks it up could settle
        # on THEIR values.  Rare path: a device-wide wait is fine.
        try:
            torch.cuda.current_stream(dev).synchronize()
        except Exception:
            pass
        with ctx.lock:
            t.status = _fused._Ticket.FAILED
            ctx.free.append(slot)
        raise
    with ctx.lock:
        ctx.pending.append(t)
    _fused._settle(t, True)                  # one C-side wait on the slot; learns capacity and depth-key bits
    if t.status == _fused._Ticket.FAILED:
        if t.patches >= 2**31:
            raise RuntimeError("splat: %d tile patches overflow int32 indexing" % t.patches)
        if t.hint < 32 and t.need > t.hint:  # stale depth-key hint: everything again (the stage is idempotent)
            gsid = render_exact(redo=True)[1]
        else:
            gsid = draw_exact(t.patches, redo=True)     # more patches than ever before
        return [image, contrib, final_tau, ranges, gsid], records(gsid)
    gsid = gsid_full[:t.patches]
    if masks:
        lists[0] = walked_full
    return [image, contrib, final_tau, ranges, gsid], records(gsid)


def splatB(height, width, us, cinv2ds, alphas, depths, colors, contrib, final_tau, patch_range_per_tile,
           gsid_per_patch, dloss_dgammas, areas=None, records=None):
    """-> [dloss_dus[N,1,2], dloss_dcinv2ds[N,1,3], dloss_dalphas[N,1,1], dloss_dcolors[N,1,3]].
    Reference: ext.cpp:20-32, gausplat.cu:114-159, kernel.cu:809-950.  ``areas``
    is an extension needed only under the forward_cpu policy (pixel boxes); ``records`` the handle of
    ``splat_with_records`` (extension; ignored unless it still describes these tensors)."""
    height, width = int(height), int(width)
    us = _chk(us, "us", torch.float32, (None, 2))
    n = us.shape[0]
    cinv2ds = _chk(cinv2ds, "cinv2ds", torch.float32, (n, 3))
    alphas = _alphas(alphas, n)
    colors = _chk(colors, "colors", torch.float32, (n, 3))
    contrib = _chk(contrib, "contrib", torch.int32, (height, width))
    final_tau = _chk(final_tau, "final_tau", torch.float32, (height, width))
    ranges = _chk(patch_range_per_tile, "patch_range_per_tile", torch.int32, (_tiles(width, height), 2))
    gsid = _chk(gsid_per_patch, "gsid_per_patch", torch.int32, (None,))
    dl = _chk(dloss_dgammas, "dloss_dgammas", torch.float32, (3, height, width))
    pol = _pol()
    if pol.footprint == 1:
        if areas is None:
            raise ValueError("splatB under the forward_cpu policy needs areas= (pixel boxes)")
        areas = _chk(areas, "areas", torch.int32, (n, 2))
    lib = _lib_on(us)
    dev = us.device
    d_us = torch.empty((n, 1, 2), dtype=torch.float32, device=dev)               # fully written by the callee
    d_cinv = torch.empty((n, 1, 3), dtype=torch.float32, device=dev)
    d_alpha = torch.empty((n, 1, 1), dtype=torch.float32, device=dev)
    d_color = torch.empty((n, 1, 3), dtype=torch.float32, device=dev)
    ws_bytes = lib.egs_splat_bwd_ws_bytes(n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    st = _stream()
    rec, order, gpack, walked = (None, None, None, None)
    seg, rebuild, seg_hint = None, 0, None
    if n > 0 and pol.footprint != 1:   # (the pixel-box policy's records also depend on `areas`, which splat mutates)
        from . import fused as _fused
        use_seg, seg_hint = _fused._seg_decision(_fused._ctx(dev), lib, (n, width, height), pol)
        if records is not None:        # explicit handle: trusted if the signatures still match
            rec, order, gpack = _take_records(records, dev, st, (us, cinv2ds, alphas, colors), width, height)
            if rec is not None:        # ... and the list with block masks, if gsid / ranges are that splat's own pair
                walked = _walked_lists(records, gsid, ranges)
                # ... and the segment states of that draw, if it split its long lists (valid for that very list pair)
                sig = _memo_sig((gsid, ranges))
                if records.seg is not None and sig is not None and sig == records.pair_sig:
                    seg = records.seg
        elif _memo_enabled:            # public pair: what the last splat of this stream kept, validated by CONTENT
            h = _splat_memo.get((dev.index, int(st.value or 0)))
            npatch = gsid.shape[0]
            if (h is not None and h.tensors is None and h.lists is not None and h.n == n and h.width == width
                    and h.height == height and h.policy == _policy_name and h.npatch == npatch > 0   # (the kept
                    # list is valid for exactly that many entries: a capacity-sized buffer holds garbage behind them)
                    and pol.alpha_skip > 0 and (gsid.data_ptr() & 15) == 0):
                rec = torch.empty((n, 12), dtype=torch.float32, device=dev)
                stamp_b = torch.empty(lib.egs_pair_stamp_words(n), dtype=torch.int32, device=dev)
                _lib.check(lib.egs_pack_records_validate(n, width, height, _ptr(us), _ptr(cinv2ds), _ptr(alphas),
                                                         _ptr(colors), C.byref(pol), _ptr(rec), _ptr(h.stamp),
                                                         _ptr(stamp_b), npatch, _ptr(h.lists), _ptr(gsid), st))
                walked, order = h.lists, h.order
                gpack, h.gpack = h.gpack, None
    kept_states = False
    if n > 0 and pol.footprint != 1:
        if seg is None and records is None and _pair_states_enabled and gsid.shape[0] > 0:
            # the public pair: the states the splat of exactly these tensors left (same four outputs by memory and
            # version -- held alive by the entry --, same inputs by memory and version, same policy and size)
            e = _pair_states.get((dev.index, int(st.value or 0)))
            if e is not None and not (e["width"] == width and e["height"] == height and e["policy"] == _policy_name):
                e = None
            if e is not None and "snap" in e:
                # CONTENT mode: differentiate the snapshot of that splat from its own states at once (self-consistent
                # whatever the caller did to the tensors since), compare what was handed in with the snapshot on the
                # device, look at the verdict once everything is enqueued (the GPU is busy with the backward kernels while
                # the host waits for one byte) and redo from the handed tensors only if a value differs
                mine = (us, cinv2ds, alphas, colors, contrib, final_tau, ranges, gsid)
                if all(a.shape == b.shape and a.dtype == b.dtype for a, b in zip(mine, e["views"])):
                    differ = torch.zeros(1, dtype=torch.int32, device=dev)
                    for a, b in zip(mine, e["views"]):          # (one pass over each byte: egs_words_differ)
                        _lib.check(lib.egs_words_differ(_ptr(a), _ptr(b), a.numel(), _ptr(differ), st))
                    e["flag"].copy_(differ, non_blocking=True)
                    verdict = torch.cuda.Event()
                    verdict.record(torch.cuda.current_stream(dev))
                    v = e["views"]
                    lists_ok = e["lists"] is not None and e["lists"].shape[0] >= gsid.shape[0]
                    _lib.check(lib.egs_splat_bwd_seg(n, gsid.shape[0], width, height, _ptr(v[0]), _ptr(v[1]), _ptr(v[2]),
                                                     _ptr(v[3]), None, C.byref(pol), _ptr(v[4]), _ptr(v[5]), _ptr(v[6]),
                                                     _ptr(e["lists"] if lists_ok else v[7]), _ptr(dl), _ptr(ws), ws_bytes,
                                                     None, None, _ptr(d_us), _ptr(d_cinv), _ptr(d_alpha), _ptr(d_color),
                                                     2 if lists_ok else 0, _ptr(e["seg"]), e["seg"].numel(), 0,
                                                     seg_hint, st))
                    verdict.synchronize()
                    if not bool(e["flag"][0]):
                        _last_splatB.update(segments=True, rebuilt=False, kept_states=True)
                        return [d_us, d_cinv, d_alpha, d_color]
                    _pair_states.pop((dev.index, int(st.value or 0)), None)     # (not this call's tensors: again, below)
            elif (e is not None and _memo_sig((contrib, final_tau, ranges, gsid)) == e["sig"]
                    and _memo_sig(e["outs"]) == e["sig"] and _memo_sig((us, cinv2ds, alphas, colors)) == e["in_sig"]):
                seg, kept_states = e["seg"], True
                if walked is None and e["lists"] is not None and e["lists"].shape[0] >= gsid.shape[0]:
                    walked = e["lists"]          # the same entries with their exact block masks
        # one entry point for every combination of what this call was handed (records / order / cleared gradient
        # records or nothing; the forward's segment states, or none: with long walks on record they are REBUILT from
        # contrib -- a forward-draw's worth of work that removes the serial tail of one wave per tile)
        if seg is None and use_seg and gsid.shape[0] > 0:
            seg = torch.empty(lib.egs_seg_rebuild_ws_bytes(gsid.shape[0], width, height), dtype=torch.uint8, device=dev)
            rebuild = 1
        _lib.check(lib.egs_splat_bwd_seg(n, gsid.shape[0], width, height, _ptr(us), _ptr(cinv2ds), _ptr(alphas),
                                         _ptr(colors), _ptr(rec), C.byref(pol), _ptr(contrib), _ptr(final_tau),
                                         _ptr(ranges), _ptr(gsid if walked is None else walked), _ptr(dl), _ptr(ws),
                                         ws_bytes, _ptr(order), _ptr(gpack), _ptr(d_us), _ptr(d_cinv), _ptr(d_alpha),
                                         _ptr(d_color), 0 if walked is None else 2, _ptr(seg),
                                         seg.numel() if seg is not None else 0, rebuild, seg_hint, st))
        _last_splatB.update(segments=seg is not None, rebuilt=bool(rebuild), kept_states=kept_states)
    elif rec is not None:   # (unreachable: records are only taken under the tile-footprint policies)
        raise AssertionError
    else:
        _lib.check(lib.egs_splat_bwd(n, gsid.shape[0], width, height, _ptr(us), _ptr(cinv2ds), _ptr(alphas),
                                     _ptr(colors), _ptr(areas), C.byref(pol), _ptr(contrib), _ptr(final_tau),
                                     _ptr(ranges), _ptr(gsid), _ptr(dl), _ptr(ws), ws_bytes, _ptr(d_us), _ptr(d_cinv),
                                     _ptr(d_alpha), _ptr(d_color), st))
    return [d_us, d_cinv, d_alpha, d_color]


# ------------------------------------------------------------------ fused chain rule (extension)
def chain_rule(dloss_dus, dloss_dcinv2ds, dloss_dcolors, Rcw, dcinv2d_dcov2ds, dcov2d_dcov3ds, dcov3d_drots,
               dcov3d_dscales, dcolor_dshs, du_dpcs, dcov2d_dpcs, dcolor_dpws):
    """One kernel for the nine batched matmuls of GSFunction.backward
    (gsmodel.py:71-85 == backward_cpu.py:476-482).
    -> (dloss_dpws[N,3], dloss_dshs[N,K], dloss_dscales[N,3], dloss_drots[N,4])."""
    n = dloss_dus.shape[0]
    g_us = _chk(dloss_dus.reshape(n, 2), "dloss_dus", torch.float32, (n, 2))
    g_ci = _chk(dloss_dcinv2ds.reshape(n, 3), "dloss_dcinv2ds", torch.float32, (n, 3))
    g_co = _chk(dloss_dcolors.reshape(n, 3), "dloss_dcolors", torch.float32, (n, 3))
    Rcw = _chk(Rcw, "Rcw", torch.float32, (3, 3))
    nc = dcolor_dshs.shape[-1]
    Js = [_chk(dcinv2d_dcov2ds, "dcinv2d_dcov2ds", torch.float32, (n, 3, 3)),
          _chk(dcov2d_dcov3ds, "dcov2d_dcov3ds", torch.float32, (n, 3, 6)),
          _chk(dcov3d_drots, "dcov3d_drots", torch.float32, (n, 6, 4)),
          _chk(dcov3d_dscales, "dcov3d_dscales", torch.float32, (n, 6, 3)),
          _chk(dcolor_dshs, "dcolor_dshs", torch.float32, (n, 1, nc)),
          _chk(du_dpcs, "du_dpcs", torch.float32, (n, 2, 3)),
          _chk(dcov2d_dpcs, "dcov2d_dpcs", torch.float32, (n, 3, 3)),
          _chk(dcolor_dpws, "dcolor_dpws", torch.float32, (n, 3, 3))]
    lib = _lib_on(g_us)
    dev = g_us.device
    dpws = torch.empty((n, 3), dtype=torch.float32, device=dev)
    dshs = torch.empty((n, 3 * nc), dtype=torch.float32, device=dev)
    dscales = torch.empty((n, 3), dtype=torch.float32, device=dev)
    drots = torch.empty((n, 4), dtype=torch.float32, device=dev)
    _lib.check(lib.egs_chain_rule(n, 3 * nc, _ptr(g_us), _ptr(g_ci), _ptr(g_co), _ptr(Rcw),
                                  *[_ptr(j) for j in Js], _ptr(dpws), _ptr(dshs), _ptr(dscales), _ptr(drots),
                                  _stream()))
    return dpws, dshs, dscales, drots
