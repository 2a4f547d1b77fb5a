"""Autograd boundary: the counterpart of the reference's ``GSFunction``
(gsplat/gsmodel.py:6-93) on top of the MI355X op surface.

Identical inputs ``(pws, shs, alphas[N,1], scales, rots, us, cam)``, outputs
``(image[3,H,W], depths > 0.2)`` and gradient tuple order (gsmodel.py:87-93).
``GSFunction.mode`` selects how the same function is evaluated:

* ``"fused"`` (default) -- easygaussiansplatting_amd.fused: one preprocess kernel +
  splat forward; splatB's draw pass + one Jacobian-free chain-rule kernel backward
  (three C-ABI calls per step, no Jacobians in HBM);
* ``"ops"``  -- the reference's structure: six op calls with ``calc_J=True``, 17
  tensors saved, ``splatB`` + the chain-rule kernel over the stored Jacobians
  (``gsplatcu.chain_rule``; tests compare it with the batched-matmul spelling of gsmodel.py:71-85).
"""
from __future__ import annotations

import dataclasses

import torch

from . import fused as _fused
from . import gsplatcu as gsc


class Camera:
    """Device-side camera, field names of reference gausplat_dataset.py:14-26."""

    def __init__(self, width, height, fx, fy, cx, cy, Rcw, tcw, device="cuda", id=0, path=""):
        self.id = id
        self.width = int(width)
        self.height = int(height)
        self.fx, self.fy, self.cx, self.cy = float(fx), float(fy), float(cx), float(cy)
        self.Rcw = torch.as_tensor(Rcw, dtype=torch.float32).to(device).contiguous()
        self.tcw = torch.as_tensor(tcw, dtype=torch.float32).to(device).contiguous()
        self.twc = (-torch.linalg.inv(self.Rcw.double().cpu()) @ self.tcw.double().cpu()).float().to(device)
        self.path = path

    @staticmethod
    def from_scene(cam, device="cuda"):
        return Camera(cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy, cam.Rcw, cam.tcw, device)


@dataclasses.dataclass(frozen=True)
class RenderOptions:
    """How ONE ``GSFunction.apply`` / ``GSRawFunction.apply`` call is evaluated -- carried by the autograd node, so two
    trainers (or a trainer and a viewer) in one process never share a switch.  Passed as the optional last argument of
    ``apply``; without it the call follows the process-wide defaults (``GSFunction.mode`` / ``ops_use_records`` and the
    ``fused.accumulate_in_kernel()`` / ``FactoredShGrad.attach()`` / ``ChunkedExchange.attach()`` blocks), which stay
    for callers that cannot change the seven-argument call of gsmodel.py:185-212."""
    mode: str = "fused"             # "fused": the fused kernels; "ops": the reference's seven-op structure
    ops_use_records: bool = True    # mode "ops": hand the forward's packed records to the backward's splatB
    accumulate: bool = False        # backward ADDS this view's gradients to the leaves' .grad inside the chain-rule kernel
    sh_sink: object = None          # dist_views.FactoredShGrad: the SH gradient of this view stays dL/dcolour [N,3]
    exchange: object = None         # dist_views.ChunkedExchange: all-reduce the gradient chunks from inside backward

    def __post_init__(self):
        if self.mode not in ("fused", "ops"):
            raise ValueError("RenderOptions.mode must be 'fused' or 'ops', got %r" % (self.mode,))
        if self.sh_sink is not None and self.exchange is not None:
            raise ValueError("RenderOptions: sh_sink and exchange exclude each other")


class GSFunction(torch.autograd.Function):
    # process-wide defaults of calls WITHOUT a RenderOptions argument
    mode = "fused"
    # mode "ops": hand the forward's packed records / masked list to the backward's splatB (gsplatcu.SplatRecords).
    # False = the plain public pair, what an UNMODIFIED reference GSFunction (gsmodel.py:6-93) gets by default.
    ops_use_records = True

    @staticmethod
    def forward(ctx, pws, shs, alphas, scales, rots, us, cam, opts=None):
        ctx.opts = opts            # None: the process-wide defaults, looked up where they are needed
        # always the maximal tuple (autograd drops surplus trailing Nones): apply(..., cam, None) passes `opts` explicitly
        ctx.n_inputs = 8
        ctx.mode = GSFunction.mode if opts is None else opts.mode
        use_records = GSFunction.ops_use_records if opts is None else opts.ops_use_records
        # the mask output never carries a gradient: do not let autograd zero-fill one per step
        ctx.set_materialize_grads(False)
        if ctx.mode == "fused":
            image, mask, state = _fused.forward(pws, shs, alphas, scales, rots, cam, need_grad=True)
            ctx.cam = cam
            ctx.state = state
            ctx.save_for_backward(pws, shs, alphas, scales, rots)
            ctx.mark_non_differentiable(mask)
            return image, mask
        # forward.md steps 1-5 == gsmodel.py:21-39
        us, pcs, depths, du_dpcs = gsc.project(pws, cam.Rcw, cam.tcw, cam.fx, cam.fy, cam.cx, cam.cy, True)
        cov3ds, dcov3d_drots, dcov3d_dscales = gsc.computeCov3D(rots, scales, depths, True)
        cov2ds, dcov2d_dcov3ds, dcov2d_dpcs = gsc.computeCov2D(
            cov3ds, pcs, cam.Rcw, depths, cam.fx, cam.fy, cam.width, cam.height, True)
        colors, dcolor_dshs, dcolor_dpws = gsc.sh2Color(shs, pws, cam.twc, True)
        cinv2ds, areas, dcinv2d_dcov2ds = gsc.inverseCov2D(cov2ds, depths, True)
        # us / cinv2ds / colors are this node's own intermediates and never leave it: the packed records of the forward
        # draw stay valid for the backward draw (gsplatcu.SplatRecords; `alphas` is checked by version in splatB)
        if use_records:
            (image, contrib, final_tau, patch_range_per_tile, gsid_per_patch), ctx.records = gsc.splat_with_records(
                cam.height, cam.width, us, cinv2ds, alphas, depths, colors, areas)
        else:
            image, contrib, final_tau, patch_range_per_tile, gsid_per_patch = gsc.splat(
                cam.height, cam.width, us, cinv2ds, alphas, depths, colors, areas)
            ctx.records = None
        ctx.cam = cam
        ctx.save_for_backward(us, cinv2ds, alphas, depths, colors, contrib, final_tau, patch_range_per_tile,
                              gsid_per_patch, dcinv2d_dcov2ds, dcov2d_dcov3ds, dcov3d_drots, dcov3d_dscales,
                              dcolor_dshs, du_dpcs, dcov2d_dpcs, dcolor_dpws)
        # depths > 0.2 after the in-place culling of inverseCov2D / splat (gsmodel.py:50): the packing kernel of splat
        # left it in the handle; otherwise one compare kernel
        mask = ctx.records.visible if (ctx.records is not None and ctx.records.visible is not None) else depths > 0.2
        ctx.mark_non_differentiable(mask)
        return image, mask

    @staticmethod
    def backward(ctx, dloss_dgammas, _):
        cam = ctx.cam
        pad = (None,) * (ctx.n_inputs - 6)      # cam (and the options)
        if dloss_dgammas is None:  # the image did not take part in the loss
            return (None,) * ctx.n_inputs
        if ctx.mode == "fused":
            pws, shs, alphas, scales, rots = ctx.saved_tensors
            o = ctx.opts
            # a training step that keeps its SH gradient factored (dist_views.FactoredShGrad): this view leaves
            # dL/dcolour [N,3] in the sink, autograd gets None for shs, the other four go on as usual
            sink = _fused.sh_sink_for(ctx, 5, (shs,), None if o is None else (o.sh_sink, o.exchange))
            leaves = (pws, alphas, scales, rots) if sink is not None else (pws, shs, alphas, scales, rots)
            acc = _fused.accumulation_targets(leaves, ctx, 5, None if o is None else (o.accumulate, o.exchange))
            dpws, dshs, dalphas, dscales, drots, dus = _fused.backward(
                pws, shs, alphas, scales, rots, cam, ctx.state, dloss_dgammas.contiguous(), accumulate=acc,
                sh_sink=sink, exchange=(_fused.DEFAULT if o is None else o.exchange))
            if acc is not None:      # added to the leaves' .grad inside the kernel: nothing for autograd to accumulate
                return (None, None, None, None, None, dus) + pad
            return (dpws, dshs, dalphas, dscales, drots, dus) + pad
        (us, cinv2ds, alphas, depths, colors, contrib, final_tau, patch_range_per_tile, gsid_per_patch,
         dcinv2d_dcov2ds, dcov2d_dcov3ds, dcov3d_drots, dcov3d_dscales, dcolor_dshs, du_dpcs, dcov2d_dpcs,
         dcolor_dpws) = ctx.saved_tensors
        dloss_dus, dloss_dcinv2ds, dloss_dalphas, dloss_dcolors = gsc.splatB(
            cam.height, cam.width, us, cinv2ds, alphas, depths, colors, contrib, final_tau,
            patch_range_per_tile, gsid_per_patch, dloss_dgammas.contiguous(), records=ctx.records)
        n = us.shape[0]
        dloss_dpws, dloss_dshs, dloss_dscales, dloss_drots = gsc.chain_rule(
            dloss_dus, dloss_dcinv2ds, dloss_dcolors, cam.Rcw, dcinv2d_dcov2ds, dcov2d_dcov3ds,
            dcov3d_drots, dcov3d_dscales, dcolor_dshs, du_dpcs, dcov2d_dpcs, dcolor_dpws)
        return (dloss_dpws, dloss_dshs, dloss_dalphas.reshape(n, 1), dloss_dscales, dloss_drots,
                dloss_dus.reshape(n, 2)) + pad


class GSRawFunction(torch.autograd.Function):
    """``GSModel.forward`` (gsmodel.py:185-212) as ONE autograd node on the optimizer's tensors:
    inputs ``(pws, low_shs, high_shs, alphas_raw, scales_raw, rots_raw, us, cam)`` -- the argument order
    of ``GSModel.forward`` -- outputs ``(image, depths > 0.2)``.  The activations (sigmoid, exp,
    normalize, cat; gsplat/utils.py:121-150) and their derivatives run inside the fused kernels."""

    @staticmethod
    def forward(ctx, pws, low_shs, high_shs, alphas_raw, scales_raw, rots_raw, us, cam, opts=None):
        ctx.opts = opts            # (``mode`` does not apply: this node IS the fused path)
        ctx.n_inputs = 9     # (as GSFunction: the maximal tuple)
        ctx.set_materialize_grads(False)
        image, mask, state = _fused.forward(pws, low_shs, alphas_raw, scales_raw, rots_raw, cam, high_shs=high_shs,
                                            need_grad=True)
        ctx.cam = cam
        ctx.state = state
        ctx.save_for_backward(pws, low_shs, high_shs, alphas_raw, scales_raw, rots_raw)
        ctx.mark_non_differentiable(mask)
        return image, mask

    @staticmethod
    def backward(ctx, dloss_dgammas, _):
        if dloss_dgammas is None:
            return (None,) * ctx.n_inputs
        pad = (None,) * (ctx.n_inputs - 7)
        o = ctx.opts
        pws, low_shs, high_shs, alphas_raw, scales_raw, rots_raw = ctx.saved_tensors
        sink = _fused.sh_sink_for(ctx, 6, (low_shs, high_shs), None if o is None else (o.sh_sink, o.exchange))
        leaves = (pws, alphas_raw, scales_raw, rots_raw) if sink is not None else \
            (pws, low_shs, high_shs, alphas_raw, scales_raw, rots_raw)
        acc = _fused.accumulation_targets(leaves, ctx, 6, None if o is None else (o.accumulate, o.exchange))
        dpws, dlow, dhigh, dalphas, dscales, drots, dus = _fused.backward(
            pws, low_shs, alphas_raw, scales_raw, rots_raw, ctx.cam, ctx.state, dloss_dgammas.contiguous(),
            high_shs=high_shs, accumulate=acc, sh_sink=sink, exchange=(_fused.DEFAULT if o is None else o.exchange))
        if acc is not None:
            return (None, None, None, None, None, None, dus) + pad
        return (dpws, dlow, dhigh, dalphas, dscales, drots, dus) + pad


def render(pws, shs, alphas, scales, rots, cam, calc_J=False):
    """Inference path of the reference's forward_gpu.py:47-60 (six op calls)."""
    us, pcs, depths = gsc.project(pws, cam.Rcw, cam.tcw, cam.fx, cam.fy, cam.cx, cam.cy, False)
    cov3ds = gsc.computeCov3D(rots, scales, depths, False)[0]
    cov2ds = gsc.computeCov2D(cov3ds, pcs, cam.Rcw, depths, cam.fx, cam.fy, cam.width, cam.height, False)[0]
    colors = gsc.sh2Color(shs, pws, cam.twc, False)[0]
    cinv2ds, areas = gsc.inverseCov2D(cov2ds, depths, False)
    return gsc.splat(cam.height, cam.width, us, cinv2ds, alphas, depths, colors, areas)
