#!/usr/bin/env python3
"""bench.py -- rendered Mpix/s, forward+backward, 1920x1080, 1 M Gaussians (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the rasterizer hot path over one batch of synthetic input, exactly as a training
step drives it (reference gsplat/gsmodel.py:6-93): ``GSFunction`` forward + backward producing the 59
parameter-gradient floats per Gaussian.  The default ``--mode fused`` evaluates it with the fused kernels
(one preprocess kernel that also bins, sort, draw; draw-backward, one Jacobian-free chain-rule kernel);
``--mode ops`` is the reference's seven-op structure (six ops with calc_J=True, splat, splatB, chain rule) --
its time is also reported as ``ops_ms_per_step`` of the default run.  The step validates its patch count the
way ``Trainer.step`` does (``fused.deferred()`` + ``commit()``): nothing is skipped inside the timed region.
With N > 1 every rank renders its own camera view of the same scene (one view per GPU, weak scaling) and the
step ends with an RCCL all-reduce of the parameter gradients (236 MB at 1 M Gaussians); `value` is the
whole-job rate: N * W*H / t_step.

Order of a run: ``--ramp-steps`` (default 400, ~0.4 s) untimed steps that take the GPU off its idle clocks, the W warm-up
steps, an untimed pre-pass with every launch bracketed by HIP events (per-kernel table), then EXACTLY K timed
steps between barrier + synchronize pairs -- back to back, the garbage collector parked, so that the timed
region sees the steady state of a training run and not the clock ramp (measured: 20 steps timed cold 1.03-1.04
ms/step, the same 20 steps after the ramp: see DESIGN 5).

Rank 0 prints ONE JSON line carrying, besides the contract fields,
  gpu_busy_ms_per_step -- sum of the HIP-event durations of all kernels of a step (untimed pre-pass), next to
                  the wall-clock ms_per_step: a ratio above 1.03 means the host, not the GPU, set the pace
                  (a warning goes to stderr);
  fwd_loss_bwd, train_step, epoch_pattern, uhd_3840x2160 -- the training step (loss, Adam), train.py's access pattern
                  (shuffled cameras, densify, reset_alpha) and one 4K render: extra legs outside the timed region;
  roofline     -- the dominant kernel of the timed region (by HIP-event time, measured on the stream it is
                  launched on), algorithmic bytes per launch / average launch duration vs the 8 TB/s HBM peak
                  and vs a device-to-device copy timed on this box (peak_measured);
  cpu_baseline -- the reference-equivalent CPU path (oracle/gs_oracle.py policy A == forward_cpu.py, single
                  thread) timed on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from tools.benchlib import (HBM_PEAK_GBS, VALU_BOUND, XGMI_LINK_GBS, algorithmic_bytes, cpu_baseline,  # noqa: E402,F401
                            epoch_pattern_leg, kernel_source_hash, parse_report, relaunch_command, scene_leg,
                            train_legs, uhd_leg)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--sh-dim", type=int, default=48)
    ap.add_argument("--ramp-steps", type=int, default=400,
                    help="untimed render steps BEFORE the warm-up steps (~0.15 s): the GPU leaves its idle clocks "
                         "(20 steps measured cold are ~7 %% slower than the steady state a training run sees); "
                         "a count, not a duration, so that all ranks issue the same collectives")
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="Gaussians in the cpu_baseline sample (0 = skip; default: the whole scene on a host with >= 64 "
                         "cores -- about 25 s of single-threaded NumPy -- else the first 250 000, extrapolated)")
    ap.add_argument("--views-per-rank", type=int, default=1,
                    help="camera views every rank renders per step (forward + backward each, gradients accumulated; "
                         "ONE gradient exchange per step): 8 on one GPU = BASELINE configs[3]'s eight ring views; on N "
                         "GPUs it amortises the 236-MB all-reduce over V renders")
    ap.add_argument("--view-streams", type=int, default=4,
                    help="with --views-per-rank > 1: HIP streams the views of a rank are dealt to (1 = one after the "
                         "other on the caller's stream); every stream accumulates its own gradient buffer, added at the "
                         "end of the step")
    ap.add_argument("--overlap-exchange", action="store_true",
                    help="several ranks, one view each: start the all-reduce of the gradient chunks inside the backward "
                         "pass (dist_views.ChunkedExchange) instead of one flat all-reduce after it")
    ap.add_argument("--factored-sh", default="auto", choices=["auto", "on", "off"],
                    help="keep the SH gradient of a step factored (dist_views.FactoredShGrad): a view leaves dL/dcolour "
                         "[N,3], the ranks all-gather 12 B per Gaussian and view instead of all-reducing the 192-B rows, "
                         "one kernel forms the rows per step.  auto: when it moves fewer bytes "
                         "(dist_views.factored_exchange_pays)")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-ops", action="store_true", help="skip the extra seven-op (--mode ops) timing")
    ap.add_argument("--no-ring8", action="store_true", help="skip the extra eight-ring-views step (configs[3])")
    ap.add_argument("--scene", default="iid", choices=["iid", "skewed", "skewed_reset"],
                    help="the synthetic scene of the timed region: iid = BASELINE's (scene.big_scene, the headline); skewed = "
                         "scene.skewed_scene (1.5 M heavy-tailed Gaussians, lists up to ~24 k); skewed_reset = the same "
                         "right after reset_alpha (every opacity <= 0.01: nothing saturates)")
    ap.add_argument("--no-skewed", action="store_true", help="skip the extra legs on the two skewed scenes")
    ap.add_argument("--no-train", action="store_true",
                    help="skip the training-step legs (fwd_loss_bwd, train_step, epoch_pattern) of the default run")
    ap.add_argument("--no-uhd", action="store_true", help="skip the extra 3840x2160 leg")
    ap.add_argument("--extras", action="store_true",
                    help="also time render+loss+backward and the whole optimizer step (other dL/dimage, so their "
                         "kernel launches would blur a rocprofv3 summary of the headline step)")
    ap.add_argument("--mode", default="fused", choices=["fused", "ops"],
                    help="GSFunction evaluation: fused kernels (default) or the reference's 7-op structure")
    ap.add_argument("--immediate", action="store_true",
                    help="fused mode: validate the patch count inside forward (one host wait per render) instead of "
                         "at the step's commit()")
    a = ap.parse_args()
    cmd = relaunch_command(a.gpus, os.environ, sys.argv[1:])
    if cmd is not None:       # `python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU)
        sys.stdout.flush()
        os.execv(cmd[0], cmd)

    import torch
    import torch.distributed as dist
    from easygaussiansplatting_amd import _lib, scene as S
    from easygaussiansplatting_amd import gsplatcu as gsc
    from easygaussiansplatting_amd.function import Camera, GSFunction, RenderOptions, render

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with `python -m torch.distributed.run "
                         "--nproc-per-node %d bench.py --gpus %d ...` (or plain `python bench.py --gpus %d`, which "
                         "re-executes itself that way)" % (a.gpus, world, a.gpus, a.gpus, a.gpus))
    # EGS_BENCH_REHEARSAL=1: every rank on cuda:0, collectives over gloo (RCCL refuses two ranks on one device) -- the
    # N > 1 code path end to end on a one-GPU box.  The numbers mean nothing (the ranks share the GPU, the exchange is
    # staged through the host) and the line says so ("rehearsal": true); what it checks is that the path runs.
    rehearsal = os.environ.get("EGS_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # EGS_FORCE_EXCHANGE=1 runs the RCCL gradient exchange even with one rank (a 1-GPU box can then
    # exercise exactly the code the multi-GPU runs execute)
    exchange = world > 1 or os.environ.get("EGS_FORCE_EXCHANGE") == "1"
    if exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm

    lib = _lib.load()
    gsc.set_policy("gsplatcu")
    from easygaussiansplatting_amd import dist_views as DV
    from easygaussiansplatting_amd import fused as fused_path

    def forward_only():
        with torch.no_grad():
            d = {k: v.detach() for k, v in params.items()}
            if a.mode == "fused":
                img, _, st = fused_path.forward(d["pws"], d["shs"], d["alphas"], d["scales"], d["rots"], cam)
                return img, st.ranges, st.gsid
            out = render(d["pws"], d["shs"], d["alphas"], d["scales"], d["rots"], cam)
            return out[0], out[3], out[4]
    if a.scene == "iid":
        sc = S.big_scene(a.gaussians, a.width, a.height, a.sh_dim)
    else:
        sc = S.skewed_scene(width=a.width, height=a.height, sh_dim=a.sh_dim, reset_alpha=(a.scene == "skewed_reset"))
    V = max(1, a.views_per_rank)
    cams = S.ring_cameras(sc.cam, max(8, world * V))
    # rank r renders views r*V .. r*V+V-1 of the ring; view 0 = the BASELINE camera (one view per GPU at V = 1)
    my_cams = [Camera.from_scene(cams[(rank * V + j) % len(cams)], dev) for j in range(V)]
    cam = my_cams[0]
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
    params = dict(pws=t(sc.pws), shs=t(sc.shs), alphas=t(sc.alphas).reshape(-1, 1).clone(), scales=t(sc.scales),
                  rots=t(sc.rots))
    for p in params.values():
        p.requires_grad_(True)
    us0 = torch.zeros((sc.n, 2), device=dev, requires_grad=True)   # gsmodel.py:198-199
    HW = a.width * a.height
    dl = torch.from_numpy(S.normal(1, 77, (3, a.height, a.width)).astype(np.float32)).to(dev) / (3 * HW)
    if V > 1:
        dl = dl / V          # the step's gradient is the MEAN over its views (train.py counterpart: Trainer.step)
    order = ("pws", "shs", "alphas", "scales", "rots")
    # the exchange may overlap the tail of the backward pass (chunked preprocess-backward, dist_views):
    # it is then launched from inside backward and only finished here
    # (only with ONE backward pass per step: a second view's `.grad +=` would race the in-flight all-reduce of the
    # same storage, dist_views.ChunkedExchange; with V > 1 the views are accumulated first and exchanged as one
    # flat buffer -- one collective for V renders)
    # --overlap-exchange (off by default since the second half of round 3): the window it overlaps is the chain-rule
    # kernel (61 us, three of four chunks), and the price is renders validated at once plus four under-filled
    # launches: on one GPU the step reads 0.99 ms with it and 0.87 without
    overlap = DV.ChunkedExchange(world) if (exchange and a.mode == "fused" and V == 1 and a.overlap_exchange) else None
    # deferred validation needs every rank to take the same decision about a redo BEFORE any collective is
    # issued; with the overlapped exchange the collectives start inside backward, so renders are then
    # validated at once (the step is exchange-bound there and the host has time to spare)
    deferred = a.mode == "fused" and not a.immediate and overlap is None
    ev_render, ev_done, ev_parts = [], [], []     # (ev_parts: render end, all-gather end, row kernel end)
    redone = [0]
    # V > 1: the views of a rank go round-robin to --view-streams HIP streams (dist_views.ViewStreams: a view is a
    # chain of dependent kernels, a quarter of it latency-bound; two views on two streams fill each other's gaps)
    n_lanes = min(V, max(1, a.view_streams)) if (a.mode == "fused" and V > 1) else 1
    vs = DV.ViewStreams([params[k] for k in order], n_lanes) if n_lanes > 1 else None
    us_lane = [us0] + [torch.zeros((sc.n, 2), device=dev, requires_grad=True) for _ in range(n_lanes - 1)]

    # The SH gradient of a step kept factored (dist_views.FactoredShGrad): 48 of the 59 gradient floats per Gaussian are
    # outer products dL/dcolour (x) basis -- a view leaves 3 floats, the exchange all-gathers them, one kernel forms
    # the rows.  auto: whenever that moves fewer bytes (any world > 1 at V = 1; several views on one GPU)
    want_fx = a.factored_sh == "on" or (a.factored_sh == "auto" and
                                        DV.factored_exchange_pays(world if exchange else 1, V, a.sh_dim))
    fx = DV.FactoredShGrad(V) if (want_fx and a.mode == "fused" and overlap is None) else None
    rest = ("pws", "alphas", "scales", "rots")

    def render_views(cams_v, vs_v, us_v, dl_v, fx_v=None):
        """forward + backward of the given views on the lanes of ``vs_v``; the parameters' .grad = sum over the views"""
        for p in params.values():
            p.grad = None
        for u in us_v:
            u.grad = None
        # what every render of the step carries (function.RenderOptions: per call, no process-wide switch): gradients
        # of further views added inside the chain-rule kernel, the SH gradient left factored in fx_v
        o = RenderOptions(mode=a.mode, accumulate=True, sh_sink=fx_v)
        if fx_v is not None:
            fx_v.begin_step(sc.n, dev)      # (rows allocated on this stream, before the lanes fork)
        vs_v.begin()
        for i, c in enumerate(cams_v):
            with vs_v.lane(i) as lv:
                image, mask = GSFunction.apply(lv[0], lv[1], lv[2], lv[3], lv[4], us_v[vs_v.lane_index(i)], c, o)
                image.backward(dl_v)
        vs_v.finish()
        if fx_v is not None and not (exchange and fx_v is fx):   # (the step's own sink under an exchange: finished
            fx_v.finish(params["pws"], params["shs"])            # there, it is a collective)
        return image

    def render_step(mode=None, records=True):
        """``mode`` / ``records``: the seven-op legs of the run (function.RenderOptions of THEIR calls; the headline's
        stay ``a.mode``)"""
        if vs is not None and mode is None:
            return render_views(my_cams, vs, us_lane, dl, fx)
        for p in params.values():
            p.grad = None
        for u in us_lane:
            u.grad = None
        # V views: forward + backward each; from the second view on the chain-rule kernel adds this view's gradients
        # to the leaves' .grad itself instead of autograd accumulating fresh tensors
        m = mode or a.mode
        sink = fx if (m == "fused") else None
        o = RenderOptions(mode=m, ops_use_records=records, accumulate=(V > 1 and m == "fused"), sh_sink=sink,
                          exchange=overlap if (m == "fused" and sink is None) else None)
        if sink is not None:
            sink.begin_step(sc.n, dev)
        if o.exchange is not None:
            o.exchange.begin_step()
        for c in my_cams:
            image, mask = GSFunction.apply(params["pws"], params["shs"], params["alphas"], params["scales"],
                                           params["rots"], us0, c, o)
            image.backward(dl)
        if sink is not None and not exchange:
            sink.finish(params["pws"], params["shs"])
        return image

    def step(timing=False):
        if timing:
            e0 = torch.cuda.Event(enable_timing=True); e0.record()
        if deferred:
            with fused_path.deferred() as d:
                image = render_step()
                bad = d.commit()       # waits for the 8-byte read-back of THIS step's binning stage only
            if bad:                    # patch list outgrew the buffers (never in steady state): exact redo
                redone[0] += 1
                image = render_step()
        else:
            image = render_step()
        if timing:
            e1 = torch.cuda.Event(enable_timing=True); e1.record()
        if exchange:  # gradient exchange: 59 floats per Gaussian, SUM then mean
            if overlap is not None and overlap.finish([params[k] for k in order]):
                pass                  # issued chunk by chunk from inside backward; now complete
            else:
                if fx is not None:        # all-gather of the views' dL/dcolour + one kernel: shs.grad = the mean rows
                    if timing:
                        eg = torch.cuda.Event(enable_timing=True); ek = torch.cuda.Event(enable_timing=True)
                        fx.finish(params["pws"], params["shs"], average=True, on_gathered=eg.record)
                        ek.record()
                        ev_parts.append((e1, eg, ek))
                    else:
                        fx.finish(params["pws"], params["shs"], average=True)
                names = rest if fx is not None else order
                flat = fused_path.flat_grad_buffer([params[k] for k in names])
                if flat is not None:      # the fused backward hands out slices of one buffer: ONE all-reduce
                    if world > 1 and dist.get_backend() == "nccl":
                        # RCCL scales inside the collective (ncclAvg): no separate 472-MB div_ pass (80 us per step
                        # at N = 1 M).  Not with ONE rank (EGS_FORCE_EXCHANGE): there AVG runs a 0.39-ms copy kernel
                        # where SUM is a no-op
                        dist.all_reduce(flat, op=dist.ReduceOp.AVG)
                    else:
                        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
                        if world > 1:
                            flat.div_(float(world))
                else:
                    grads = [params[k].grad for k in names]
                    hs = [dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True) for g in grads]
                    for h in hs:
                        h.wait()
                    torch._foreach_div_(grads, float(world))
        if timing:
            e2 = torch.cuda.Event(enable_timing=True); e2.record()
            ev_render.append((e0, e1)); ev_done.append((e1, e2))
        return image

    def sync():
        if exchange:
            dist.barrier()
        torch.cuda.synchronize()

    import ctypes
    import gc
    # the generational collector stays off from here to the end of the timed region: a collection inside a
    # ~20 ms region is a 10-50 % outlier, and a full collection BETWEEN warm-up and timing (it takes tens of ms)
    # lets the GPU fall back to its idle clocks right before the clock starts
    gc.collect()
    gc.disable()
    for _ in range(max(0, a.ramp_steps)):   # clock ramp (no effect on what a step computes)
        step()
    for _ in range(a.warmup):
        step()
    sync()
    prof = not a.no_prof

    def read_report():
        need = lib.egs_prof_report(None, 0)
        buf = ctypes.create_string_buffer(need + 16)
        lib.egs_prof_report(buf, need + 16)
        return parse_report(buf.value.decode())

    # Untimed pre-pass with EVERY launch bracketed by HIP events: per-kernel table + which kernel
    # dominates.  (Bracketing all ~30 launches serialises dispatch and costs ~0.2 ms/step, so the
    # timed region below brackets only the dominant kernel.)
    kernels, dom, gpu_busy_ms = {}, None, None
    if prof:
        pre = 3
        lib.egs_prof_set_filter(None); lib.egs_prof_reset(); lib.egs_prof_enable(1)
        for _ in range(pre):
            step()
        sync()
        lib.egs_prof_enable(0)
        rep = read_report()
        kernels = {k: {"launches_per_step": c // pre, "avg_us": round(tot / c * 1e3, 2),
                       "ms_per_step": round(tot / pre, 4)}
                   for k, (c, tot) in sorted(rep.items(), key=lambda kv: -kv[1][1])}
        gpu_busy_ms = sum(tot for _, tot in rep.values()) / pre
        dom = max(rep.items(), key=lambda kv: kv[1][1])[0]
        lib.egs_prof_set_filter(dom.encode()); lib.egs_prof_reset(); lib.egs_prof_enable(1)
        sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        image = step()
    sync()
    dt = time.perf_counter() - t0
    gc.enable()
    if prof:
        lib.egs_prof_enable(0)
    dt_t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if exchange:
        dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
    dt = float(dt_t.item())
    ms = dt / a.steps * 1e3

    roofline_rep = read_report() if prof else None   # only the dominant kernel, recorded over the timed region
    if prof:
        lib.egs_prof_set_filter(None)

    # per-rank anatomy of a step (outside the timed region): render = forward + backward kernels,
    # exchange = what the gradient all-reduce adds behind them
    exch = None
    if exchange:
        for _ in range(5):
            step(timing=True)
        sync()
        tr = float(np.mean([x.elapsed_time(y) for x, y in ev_render]))
        te = float(np.mean([x.elapsed_time(y) for x, y in ev_done]))
        mine = torch.tensor([tr, te], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        nbytes = DV.grad_exchange_bytes(sc.n)
        te_max = max(float(x[1]) for x in allr)
        form = {"form": "flat all-reduce of 59 floats per Gaussian", "link_bytes_per_rank": 2 * (world - 1) / world * nbytes,
                "direct_bytes_per_link": 2 * nbytes / world}
        if fx is not None:
            # all-gather: a rank receives (world - 1) * V rows of 12 N + 16 bytes; all-reduce of the other 11 floats
            rest_b = 4 * 11 * sc.n
            ag = (world - 1) * V * 4 * DV.FactoredShGrad.row_stride(sc.n)
            form = {"form": "SH gradient factored: all-gather of dL/dcolour (12 B per Gaussian and view) + all-reduce of "
                            "11 floats per Gaussian + one kernel forming the rows",
                    "link_bytes_per_rank": ag + 2 * (world - 1) / world * rest_b,
                    "direct_bytes_per_link": V * 4 * DV.FactoredShGrad.row_stride(sc.n) + 2 * rest_b / world}
        # the two collectives of the factored form timed apart (the first real SCALE line should decide
        # --factored-sh auto's threshold from the all-gather's and the all-reduce's own bus bandwidths, not from
        # factored_exchange_pays()'s byte count): this rank's times; bytes a rank sends + receives per collective
        parts = None
        if fx is not None and ev_parts:
            t_ag = float(np.mean([a_.elapsed_time(b_) for a_, b_, _ in ev_parts]))
            t_rows = float(np.mean([b_.elapsed_time(c_) for _, b_, c_ in ev_parts]))
            t_ar = float(np.mean([c_.elapsed_time(d_[1]) for (_, _, c_), d_ in zip(ev_parts, ev_done)]))
            ag_b = (world - 1) * V * 4 * DV.FactoredShGrad.row_stride(sc.n)
            ar_b = 2 * (world - 1) / world * 4 * 11 * sc.n
            parts = {"all_gather": {"ms": round(t_ag, 4), "link_bytes_per_rank": ag_b,
                                    "bus_GBs": round(ag_b / (t_ag * 1e-3) / 1e9, 1) if (world > 1 and t_ag > 0) else None},
                     "rows_kernel_ms": round(t_rows, 4),
                     "all_reduce": {"ms": round(t_ar, 4), "link_bytes_per_rank": ar_b,
                                    "bus_GBs": round(ar_b / (t_ar * 1e-3) / 1e9, 1) if (world > 1 and t_ar > 0) else None}}
        exch = {"bytes": nbytes, **form, "collectives": parts, "t_render_ms": [round(float(x[0]), 4) for x in allr],
                "t_exchange_ms": [round(float(x[1]), 4) for x in allr],
                "overlapped_with_backward": bool(overlap is not None and overlap.used),
                "bus_GBs": round(form["link_bytes_per_rank"] / (te_max * 1e-3) / 1e9, 1) if world > 1 else None,
                "xgmi_bound_ms": None if world == 1 else {
                    "ring_one_link": round(form["link_bytes_per_rank"] / (XGMI_LINK_GBS * 1e9) * 1e3, 3),
                    "direct_all_links": round(form["direct_bytes_per_link"] / (XGMI_LINK_GBS * 1e9) * 1e3, 3)}}

    # realised scene statistics (bytes depend on them)
    with torch.no_grad():
        _, ranges, gsid = forward_only()
        lens = (ranges[:, 1] - ranges[:, 0]).to(torch.int64)
        P_drawn = int(gsid.shape[0]); T = int(ranges.shape[0])
        # P of the reference's lists (every tile of every rect, getRects kernel.cu:82-122): what SURVEY 8(d)'s
        # algorithmic-byte formulas count.  The fused path draws footprint-culled lists (P_drawn <= P).
        P = int(render(params["pws"].detach(), params["shs"].detach(), params["alphas"].detach(),
                       params["scales"].detach(), params["rots"].detach(), cam)[4].shape[0]) if a.mode == "fused" \
            else P_drawn
        max_len = int(lens.max().item())
        pairs = int(lens.sum().item()) * 256
        # forward-only rate (BASELINE configs[1]), untimed by the headline.  The statistics above went through host
        # reads and a seven-op render: re-ramp the clocks first (as the headline does), then time.
        nf = max(3, a.steps // 2)
        for _ in range(min(max(0, a.ramp_steps), 200)):
            forward_only()
        torch.cuda.synchronize()
        tf0 = time.perf_counter()
        for _ in range(nf):
            forward_only()
        torch.cuda.synchronize()
        fwd_ms = (time.perf_counter() - tf0) / nf * 1e3
    # per-kernel algorithmic rate of the headline step (each input read once, each output written once)
    for k, row in kernels.items():
        ab = algorithmic_bytes(k, sc.n, P, T, HW, a.sh_dim, (world * V) if fx is not None else 0)
        if ab and row.get("avg_us"):
            row["algorithmic_GBs"] = round(ab / (row["avg_us"] * 1e-6) / 1e9, 1)
            row["frac_of_hbm_peak"] = round(ab / (row["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 3)

    # the unmodified-caller surface: GSFunction over the seven ops (six with calc_J=True, splat, splatB and the
    # chain-rule kernel over the stored Jacobians) -- an extra, outside the timed region
    ops_ms, ops_kernels, ops_public_ms = None, None, None
    if a.mode == "fused" and not a.no_ops and rank == 0 and world == 1:
        for _ in range(8):        # (the first calls allocate the Jacobian tensors and learn the patch capacity)
            render_step("ops")
        torch.cuda.synchronize()
        to0 = time.perf_counter()
        for _ in range(40):
            render_step("ops")
        torch.cuda.synchronize()
        ops_ms = (time.perf_counter() - to0) / 40 * 1e3 / V      # per view
        # the same step WITHOUT the records handle: the public splat / splatB pair as an unmodified reference
        # GSFunction (gsmodel.py:6-93) calls it -- splatB packs its own records and walks the plain list
        for _ in range(4):
            render_step("ops", records=False)
        torch.cuda.synchronize()
        to0 = time.perf_counter()
        for _ in range(40):
            render_step("ops", records=False)
        torch.cuda.synchronize()
        ops_public_ms = (time.perf_counter() - to0) / 40 * 1e3 / V
        if prof:      # per-kernel table of the seven-op step (event-bracketed, outside the timing above)
            lib.egs_prof_set_filter(None); lib.egs_prof_reset(); lib.egs_prof_enable(1)
            for _ in range(3):
                render_step("ops")
            torch.cuda.synchronize()
            lib.egs_prof_enable(0)
            orep = read_report()
            lib.egs_prof_reset()
            ops_kernels = {}
            for k, (c, tot) in sorted(orep.items(), key=lambda kv: -kv[1][1]):
                row = {"launches_per_step": c // (3 * V), "avg_us": round(tot / c * 1e3, 2),
                       "ms_per_step": round(tot / (3 * V), 4)}
                ab = algorithmic_bytes(k, sc.n, P, T, HW, a.sh_dim)
                if ab:
                    row["algorithmic_GBs"] = round(ab / (tot / c * 1e-3) / 1e9, 1)
                    row["frac_of_hbm_peak"] = round(ab / (tot / c * 1e-3) / 1e9 / HBM_PEAK_GBS, 3)
                ops_kernels[k] = row
        for p in params.values():
            p.grad = None
        torch.cuda.empty_cache()

    # BASELINE configs[3] on this one GPU: the eight ring views as ONE step (views on --view-streams streams, gradients
    # accumulated, no exchange) -- an extra, outside the timed region
    ring8 = None
    if a.mode == "fused" and not a.no_ring8 and rank == 0 and world == 1 and V == 1 and not a.immediate:
        cams8 = [Camera.from_scene(c, dev) for c in S.ring_cameras(sc.cam, 8)]
        lanes8 = max(1, min(8, a.view_streams))
        vs8 = DV.ViewStreams([params[k] for k in order], lanes8)
        us8 = [torch.zeros((sc.n, 2), device=dev, requires_grad=True) for _ in range(lanes8)]
        dl8 = dl / 8

        fx8 = DV.FactoredShGrad(8) if a.factored_sh != "off" else None     # eight views, one GPU: it pays

        def step8():
            with fused_path.deferred() as d8:
                render_views(cams8, vs8, us8, dl8, fx8)
                if d8.commit():          # (a view outgrew the buffers learnt so far: exact redo)
                    render_views(cams8, vs8, us8, dl8, fx8)
        for _ in range(6):
            step8()
        torch.cuda.synchronize()
        t80 = time.perf_counter()
        for _ in range(12):
            step8()
        torch.cuda.synchronize()
        ms8 = (time.perf_counter() - t80) / 12 * 1e3
        ring8 = {"views": 8, "view_streams": lanes8, "factored_sh": fx8 is not None, "ms_per_step": round(ms8, 4),
                 "Mpix/s": round(8 * HW / (ms8 * 1e-3) / 1e6, 2),
                 "what": "forward + backward of the 8 ring cameras of BASELINE configs[3] as one step on this GPU "
                         "(12 steps after 6 untimed; host clock around a synchronize)"}
        del vs8, us8
        for p in params.values():
            p.grad = None
        torch.cuda.empty_cache()

    # the same step on the heavy-tailed scenes (an extra, outside the timed region)
    skewed = None
    if (a.mode == "fused" and a.scene == "iid" and not a.no_skewed and rank == 0 and world == 1 and V == 1
            and not a.immediate and prof):
        ref = {"pairs": pairs, "k_draw_us": kernels.get("k_draw", {}).get("avg_us"),
               "k_draw_bwd_us": kernels.get("k_draw_bwd", {}).get("avg_us")}
        for p in params.values():
            p.grad = None
        torch.cuda.empty_cache()
        skewed = [scene_leg(nm, S.skewed_scene(width=a.width, height=a.height, sh_dim=a.sh_dim, reset_alpha=rs), dev, lib,
                            12, ref, train=(rs and not a.no_train))
                  for nm, rs in (("skewed", False), ("skewed_reset", True))]

    # achievable HBM bandwidth on THIS box: a device-to-device float4 copy ("confirm on the box
    # with a device-to-device copy kernel and report both")
    peak_measured, clock_mhz = None, None
    if rank == 0:
        nb = 1 << 30
        src = torch.empty(nb, dtype=torch.uint8, device=dev).fill_(1)
        dst = torch.empty(nb, dtype=torch.uint8, device=dev)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(2):
            _lib.check(lib.egs_hbm_copy_probe(ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(src.data_ptr()), nb, st))
        best = 0.0
        for _ in range(3):      # best of three batches (the clock settles during the first)
            c0 = torch.cuda.Event(enable_timing=True); c1 = torch.cuda.Event(enable_timing=True)
            c0.record()
            reps = 8
            for _ in range(reps):
                _lib.check(lib.egs_hbm_copy_probe(ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(src.data_ptr()), nb,
                                                  st))
            c1.record()
            torch.cuda.synchronize()
            best = max(best, 2.0 * nb * reps / (c0.elapsed_time(c1) * 1e-3) / 1e9)   # read + write
        peak_measured = round(best, 1)
        del src, dst
        # shader clock under full VALU load (a chain of dependent v_fma on every SIMD between the shader-clock counter and
        # the constant 100-MHz one): what 1024 SIMDs x clock means on THIS box right now
        cb = torch.zeros(8, dtype=torch.int64, device=dev)
        for _ in range(3):
            _lib.check(lib.egs_clock_probe(ctypes.c_void_p(cb.data_ptr()), 20000, st))
        torch.cuda.synchronize()
        cbh = cb.cpu().numpy()
        clock_mhz = round(float(cbh[1] - cbh[0]) / max(1.0, float(cbh[3] - cbh[2])) * 100.0, 1)

    # the whole training step (outside the timed region; VERDICT r5 #3: in the DEFAULT line): render + fused HIP L1 / SSIM
    # loss + backward, and raw parameters -> activations -> render -> loss -> backward -> Adam; --extras adds the
    # reference's structure (torch activations, torch.optim.Adam).  Then the access pattern of train.py's loop
    # (epoch_pattern) and one 4K render.
    loss_leg, train_extra, epoch_pat, uhd = None, None, None, None
    solo = a.mode == "fused" and a.scene == "iid" and rank == 0 and world == 1 and V == 1 and not a.immediate
    if solo and (a.extras or not a.no_train):
        for p in params.values():
            p.grad = None
        torch.cuda.empty_cache()
        loss_leg, train_extra = train_legs(sc, cam, dev, nf, full=a.extras)
        epoch_pat = epoch_pattern_leg(dev)
    if solo and not a.no_uhd and a.width == 1920:
        uhd = uhd_leg(dev, lib)

    roofline = None
    src_hash = kernel_source_hash()
    if prof and roofline_rep and dom in roofline_rep:
        cnt, tot = roofline_rep[dom]
        avg_s = tot / cnt * 1e-3
        ab = algorithmic_bytes(dom, sc.n, P, T, HW, a.sh_dim)
        if ab is not None:
            ach = ab / avg_s / 1e9
            # what BOUNDS the dominant kernel: the two draw kernels issue VALU instructions 80-95 % of their SIMD cycles and
            # move 1.3x / 2.5x their algorithmic bytes at 0.15 of the HBM peak -- `frac` (vs HBM, the contract's field) can
            # never move for them; `valu.frac` (below, from the counter passes under profiles/) is their roofline
            roofline = {"kernel": dom, "bound": "valu" if dom in VALU_BOUND else "hbm", "achieved": round(ach, 1),
                        "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                        "peak_measured": peak_measured, "clock_mhz": clock_mhz,
                        "frac_of_measured": None if not peak_measured else round(ach / peak_measured, 4),
                        "algorithmic_bytes_per_launch": ab, "avg_launch_us": round(avg_s * 1e6, 1),
                        "launches": cnt, "share_of_step": round(tot / a.steps / ms, 3)}
            # `achieved` / `frac` above count SURVEY 8(d)'s bytes on the REFERENCE's lists (`patches`: what the seven-op
            # surface walks); the fused path walks footprint-culled lists (`patches_drawn`): the same figure on those
            abd = algorithmic_bytes(dom, sc.n, P_drawn, T, HW, a.sh_dim)
            if abd is not None:
                roofline["on_lists_drawn"] = {"algorithmic_bytes_per_launch": abd,
                                              "achieved": round(abd / avg_s / 1e9, 1),
                                              "frac": round(abd / avg_s / 1e9 / HBM_PEAK_GBS, 4)}
            # the whole step against the HBM peak, two byte counts: SURVEY 8(d)'s formula for the seven-op surface
            # (1656 N + 316 P + 24 T + 40 HW at K = 48: Jacobians written and re-read, the reference's lists) and the
            # bytes the FUSED path's own kernels move algorithmically (no Jacobians, culled lists, sorts as 12 B per item
            # and pass)
            if a.mode == "fused" and kernels:
                K_ = a.sh_dim
                # N-terms: stages 204 + 4K, Jacobians 372 + 4K/3, splat 72, splatB 36, chain rule 460 + 4K + 4K/3 (= 1656 at K = 48)
                survey = (1144 + 8 * K_ + 8 * K_ // 3) * sc.n + 316 * P + 24 * T + 40 * HW
                own = 0
                for k_, row_ in kernels.items():
                    ab_ = algorithmic_bytes(k_, sc.n, P_drawn, T, HW, a.sh_dim, (world * V) if fx is not None else 0)
                    if ab_ is None and k_ in ("k_radix_hist", "k_radix_scatter"):
                        # per step: 2 depth passes over N items + 2 tile passes over P_drawn items; hist reads 4 B,
                        # scatter moves 16 B per item (the depth sort's last pass also gathers 16 + 4 B)
                        per = (4 if k_ == "k_radix_hist" else 16)
                        ab_ = (2 * sc.n + 2 * P_drawn) * per / max(1, row_["launches_per_step"]) + \
                            (20 * sc.n / max(1, row_["launches_per_step"]) if k_ == "k_radix_scatter" else 0)
                    if ab_:
                        own += ab_ * row_["launches_per_step"]
                roofline["step"] = {
                    "survey_8d_bytes": int(survey), "survey_8d_frac": round(survey / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "fused_own_bytes": int(own), "fused_own_frac": round(own / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "note": "survey_8d = the seven-op surface's algorithmic bytes (SURVEY 8d) over THIS step's time; "
                            "fused_own = what the fused path's kernels move algorithmically (no Jacobians, culled lists)"}
            # HBM bytes per launch and the calibrated VALU-issue utilisation come from rocprofv3 --pmc passes stored
            # under profiles/ -- quoted only when those passes ran on exactly these kernel sources
            tpath = os.path.join(REPO, "profiles", "pmc_traffic.json")
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    if (tj.get("source_hash") == src_hash and tj.get("gaussians") == sc.n and
                            tj.get("width") == a.width and dom in tj.get("kernels", {})):
                        kj = tj["kernels"][dom]
                        roofline["traffic"] = kj["hbm_bytes_per_launch"]
                        if "valu_issue_util" in kj:
                            # frac: calibrated issue cycles of the VALU instructions the SQ counters saw (2.5 / 4.3 / 8.5
                            # SIMD cycles for full / half / quarter rate, tools/lab/ubench_calib.hip; quarter-rate count from
                            # the counters, the half-rate share of the rest from the kernel's ISA, tools/valu_mix.py) over
                            # the launch's SIMD cycles -- both in the counters' clock domain; band: every other
                            # instruction full rate .. half rate
                            roofline["valu"] = {
                                "frac": kj.get("valu_frac"), "band": kj["valu_issue_util"],
                                "insts_per_launch": kj.get("valu_insts_per_launch"),
                                "quarter_rate_share": kj.get("valu_quarter_rate_share"),
                                "half_rate_share_static": kj.get("valu_half_rate_share_static"),
                                "simd_cycles_per_launch": kj.get("simd_cycles_per_launch"),
                                "launch_us_in_counter_pass": kj.get("duration_us_in_counter_pass"),
                                "gui_clock_mhz": kj.get("gui_clock_mhz"),
                                "note": "SIMD-cycle share in which the VALU port issues; the rest is scalar / LDS issue and "
                                        "stalls no other wave of the SIMD covers (DESIGN 3.4)"}
                except Exception:
                    pass

    cpu = None
    cpu_sample = a.cpu_sample
    if cpu_sample < 0:
        cpu_sample = sc.n if (os.cpu_count() or 1) >= 64 else 250_000
    if rank == 0 and world == 1 and cpu_sample > 0:
        cpu = cpu_baseline(sc, min(cpu_sample, sc.n))

    if rank == 0:
        value = world * V * HW / (ms * 1e-3) / 1e6       # every rank renders V full frames per step
        line = {
            "metric": "rendered Mpix/s fwd+bwd at 1920x1080, 1M Gaussians" if a.scene == "iid" else
                      "rendered Mpix/s fwd+bwd at %dx%d, scene.skewed_scene (%s)" % (a.width, a.height, a.scene), "value": round(value, 2),
            "unit": "Mpix/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            **({"rehearsal": "EGS_BENCH_REHEARSAL=1: all ranks on ONE GPU, collectives over gloo -- the numbers of this "
                             "line are not measurements of the multi-GPU path"} if rehearsal else {}),
            "config": {"workload": "%s: %d synthetic Gaussians, %dx%d, SH degree %d, "
                                   "forward+backward (GSFunction, mode=%s)%s"
                                   % ("1 MI355X per view" if V == 1 else
                                      "%d ring views per MI355X and step (BASELINE configs[3]'s views, gradients "
                                      "accumulated)" % V,
                                      sc.n, a.width, a.height, {3: 0, 12: 1, 27: 2, 48: 3}[a.sh_dim], a.mode,
                                      "" if world == 1 else
                                      (", RCCL all-gather of dL/dcolour (3 fp32 per Gaussian and view) + all-reduce of "
                                       "11 fp32 grads/Gaussian, SH rows formed per step" if fx is not None else
                                       ", RCCL all-reduce of 59 fp32 grads/Gaussian")),
                       "gaussians": sc.n, "width": a.width, "height": a.height, "sh_dim": a.sh_dim,
                       "views_per_step": world * V, "views_per_rank": V, "view_streams": n_lanes, "policy": "gsplatcu", "mode": a.mode,
                       "validation": "deferred (commit per step)" if deferred else "immediate",
                       "preconditioning": "%d untimed steps before the warm-up steps (clock ramp)" % max(0, a.ramp_steps),
                       "tile_dispatch": "forward: by the work measured at this camera's previous render (list "
                                        "length at first sight); backward: by the work this render measured"
                                        if fused_path.TILE_WORK_CACHE else "forward: by list length; backward: by "
                                        "the work this render measured",
                       "patches": P, "patches_drawn": P_drawn, "tiles": T, "max_list_len": max_len,
                       "pixel_gaussian_pairs": pairs,
                       "lists": "footprint-culled (fused path)" if (a.mode == "fused" and fused_path.CULL_LISTS)
                                else "reference rects"},
            "gpu_busy_ms_per_step": None if gpu_busy_ms is None else round(gpu_busy_ms, 4),
            "wall_over_gpu_busy": None if not gpu_busy_ms else round(ms / gpu_busy_ms, 4),
            "redone_steps": redone[0],
            "fwd_only": {"ms": round(fwd_ms, 4), "Mpix/s": round(HW / (fwd_ms * 1e-3) / 1e6, 2)},
            "ops_ms_per_step": None if ops_ms is None else round(ops_ms, 4),
            "ops_public_pair_ms_per_step": None if ops_public_ms is None else round(ops_public_ms, 4),
            "ops_kernels": ops_kernels,
            "ring_views_8": ring8, "skewed_scenes": skewed,
            "fwd_loss_bwd": loss_leg, "train_step": train_extra, "epoch_pattern": epoch_pat, "uhd_3840x2160": uhd,
            "exchange": exch,
            "roofline": roofline, "cpu_baseline": cpu, "kernels": kernels, "kernel_source_hash": src_hash,
        }
        if cpu:
            line["fwd_speedup_vs_cpu"] = round(line["fwd_only"]["Mpix/s"] / cpu["value"], 1)
        if gpu_busy_ms and world == 1 and ms > 1.03 * gpu_busy_ms:
            print("bench.py: WARNING: wall-clock %.4f ms/step is %.1f %% above the %.4f ms the kernels take: the "
                  "host, not the GPU, set the pace of this run" % (ms, (ms / gpu_busy_ms - 1) * 100, gpu_busy_ms),
                  file=sys.stderr)
    if exchange:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes a version banner to the C stdout buffer: push it out first so that the JSON
        # line is the LAST line of output
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
