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

import ctypes as C
import os

import torch

from . import _lib
from . import gsplatcu as _gsc
from .dist_views import flat_grad_buffer  # noqa: F401  (re-exported: the buffer is allocated here)
from .gsplatcu import _alphas, _bin_stage, _chk, _lib_on, _pol, _ptr, _stream, _tiles


ENQUEUE_AHEAD = os.environ.get("EGS_ENQUEUE_AHEAD", "1") != "0"   # knob for A/B measurements and tests
_patch_capacity = {}     # (N, W, H) -> patch-list allocation size learnt from earlier calls
_mailbox = {}            # device index -> page-locked int32[2]: landing zone of {P, max depth key}


class FusedState:
    """Tensors the backward pass needs (all produced by ``forward``)."""
    __slots__ = ("us", "depths", "cinv2ds", "colors", "areas", "rec", "contrib", "final_tau", "ranges", "gsid",
                 "width", "height")


def _split_sh(low_shs, high_shs, n):
    low = _chk(low_shs, "low_shs", torch.float32, (n, 3))
    high = _chk(high_shs, "high_shs", torch.float32, (n, None))
    K = 3 + high.shape[1]
    if K not in (3, 12, 27, 48):
        raise ValueError("low_shs + high_shs must have 3, 12, 27 or 48 columns, got %d" % K)
    return low, high, K


def forward(pws, shs, alphas, scales, rots, cam, high_shs=None):
    """-> (image[3,H,W], mask[N] bool, state).  ``cam`` carries Rcw/tcw/twc device
    tensors and fx, fy, cx, cy, width, height (reference gausplat_dataset.py:14-26).
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
    # the draw kernels (forward and backward) work from the packed records alone: us / cinv2ds / colors /
    # areas are not materialised
    S.us = S.cinv2ds = S.colors = S.areas = None
    S.depths = torch.empty((n,), dtype=f32, device=dev)
    S.rec = torch.empty((max(n, 1), 12), dtype=f32, device=dev)   # packed 2D records, reused by backward
    mask = torch.empty((n,), dtype=torch.bool, device=dev)        # depths > 0.2, written by the kernel
    ws_bin_bytes = lib.egs_splat_bin_ws_bytes(n)
    ws_bin = torch.empty(ws_bin_bytes, dtype=torch.uint8, device=dev)
    tail = lambda hint, total: (_ptr(alphas), _ptr(Rcw), _ptr(tcw), _ptr(twc), float(cam.fx), float(cam.fy),
                                float(cam.cx), float(cam.cy), W, H, pol, _ptr(S.us), _ptr(S.depths), _ptr(S.cinv2ds),
                                _ptr(S.colors), _ptr(S.areas), _ptr(S.rec), _ptr(mask), hint, _ptr(ws_bin),
                                ws_bin_bytes, _ptr(total), st)
    image = torch.empty((3, H, W), dtype=f32, device=dev)       # fully written by the draw stage
    S.contrib = torch.empty((H, W), dtype=i32, device=dev)
    S.final_tau = torch.empty((H, W), dtype=f32, device=dev)
    S.ranges = torch.empty((_tiles(W, H), 2), dtype=i32, device=dev)

    def draw_exact(patches):
        S.gsid = torch.empty(patches, dtype=i32, device=dev)
        ws_draw = torch.empty(lib.egs_splat_draw_ws_bytes(n, patches, W, H), dtype=torch.uint8, device=dev)
        _lib.check(lib.egs_splat_draw_rec(n, patches, W, H, _ptr(S.rec), pol, _ptr(ws_bin), _ptr(ws_draw),
                                          ws_draw.numel(), _ptr(image), _ptr(S.contrib), _ptr(S.final_tau),
                                          _ptr(S.ranges), _ptr(S.gsid), st))

    if raw:
        enqueue_bin = lambda hint, total: _lib.check(lib.egs_fused_forward_raw(
            n, K, _ptr(pws), _ptr(rots), _ptr(scales), _ptr(shs), _ptr(high_shs), *tail(hint, total)))
    else:
        enqueue_bin = lambda hint, total: _lib.check(lib.egs_fused_forward(
            n, K, _ptr(pws), _ptr(rots), _ptr(scales), _ptr(shs), *tail(hint, total)))

    cap = _patch_capacity.get((n, W, H), 0) if ENQUEUE_AHEAD else 0
    if cap == 0 or n == 0:
        patches = _bin_stage(enqueue_bin)            # first call for this size: synchronous read-back of P
        draw_exact(patches)
    else:
        # The draw stage is enqueued AHEAD of the read-back: buffers sized by the largest patch count seen so
        # far, the kernels take the real count from device memory, and {P, max depth key} travel to a
        # page-locked mailbox by a copy enqueued between the two stages.  The host then only polls that
        # mailbox -- the GPU never waits for it (the reference, like the seven-op path, idles around
        # cudaMemcpy(&P), gausplat.cu:67).  An overflow of the capacity or of the depth-key hint is detected
        # here, after the fact, and the affected stage is redone.
        if dev.index not in _mailbox:
            _mailbox[dev.index] = torch.zeros(2, dtype=torch.int32).pin_memory()
        box = _mailbox[dev.index]
        box.fill_(-1)                                 # sentinels: neither P nor a depth key is ever 0xFFFFFFFF
        total = torch.empty(2, dtype=i32, device=dev)
        hint = _gsc._key_bits_hint
        enqueue_bin(hint, total)
        gsid_full = torch.empty(cap, dtype=i32, device=dev)
        ws_draw = torch.empty(lib.egs_splat_draw_ws_bytes(n, cap, W, H), dtype=torch.uint8, device=dev)
        _lib.check(lib.egs_splat_draw_rec_dev(n, cap, _ptr(total), C.c_void_p(box.data_ptr()), W, H, _ptr(S.rec), pol,
                                              _ptr(ws_bin), _ptr(ws_draw), ws_draw.numel(), _ptr(image),
                                              _ptr(S.contrib), _ptr(S.final_tau), _ptr(S.ranges), _ptr(gsid_full),
                                              st))
        spins = 0
        while int(box[0]) == -1 or int(box[1]) == -1:  # arrives ~0.2 ms before the draw stage finishes
            spins += 1
            if spins > 200000:                        # (never observed) fall back to a real synchronisation
                torch.cuda.current_stream().synchronize()
                break
        patches, mk = int(box[0]) & 0xFFFFFFFF, int(box[1]) & 0xFFFFFFFF
        need = mk.bit_length()
        if patches >= 2**31:
            raise RuntimeError("splat: %d tile patches overflow int32 indexing" % patches)
        if hint < 32 and need > hint:                 # stale depth-key hint: everything again, full key width
            _gsc._key_bits_hint = 32
            patches = _bin_stage(enqueue_bin)
            draw_exact(patches)
        else:
            _gsc._key_bits_hint = min(32, need + 1)
            if patches > cap:                         # more patches than ever before: redo the draw stage
                draw_exact(patches)
            else:
                S.gsid = gsid_full[:patches]
    if n > 0:
        _patch_capacity[(n, W, H)] = max(_patch_capacity.get((n, W, H), 0), patches + patches // 32 + 4096)
    return image, mask, S


def backward(pws, shs, alphas, scales, rots, cam, S: FusedState, dloss_dgammas, high_shs=None):
    """-> (dloss_dpws[N,3], dloss_dshs[N,K], dloss_dalphas[N,1], dloss_dscales[N,3],
           dloss_drots[N,4], dloss_dus[N,2])  -- the gradient tuple of gsmodel.py:87-93.
    With ``high_shs`` (raw tensors, see ``forward``): -> (dpws, dlow_shs[N,3], dhigh_shs[N,K-3],
    dalphas_raw[N,1], dscales_raw, drots_raw, dus)."""
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
    starts, at = [], 0
    for w in widths:                      # every slice starts 16-B aligned (the kernels store dwordx4)
        starts.append(at)
        at += (n * w + 3) // 4 * 4
    flat = torch.empty(at, dtype=f32, device=dev)
    parts = [flat[a:a + n * w].view(n, w) for a, w in zip(starts, widths)]
    if raw:
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
    if raw:
        _lib.check(lib.egs_fused_backward_raw(n, K, S.gsid.shape[0], W, H, _ptr(pws), _ptr(rots), _ptr(scales),
                                              _ptr(shs), _ptr(high_shs), *mid, _ptr(dhigh), _ptr(dalphas),
                                              _ptr(dscales), _ptr(drots), _ptr(dus), _stream()))
        return dpws, dshs, dhigh, dalphas, dscales, drots, dus
    _lib.check(lib.egs_fused_backward(n, K, S.gsid.shape[0], W, H, _ptr(pws), _ptr(rots), _ptr(scales), _ptr(shs),
                                      *mid, _ptr(dalphas), _ptr(dscales), _ptr(drots), _ptr(dus), _stream()))
    return dpws, dshs, dalphas, dscales, drots, dus
