"""Helpers of bench.py (the contract line itself is assembled there): algorithmic byte counts per kernel,
the kernel-source fingerprint that ties profiles/pmc_traffic.json to the code it was measured on, the CPU baseline leg
(the oracle as the reference-equivalent CPU path: never the product), the self-launcher for N > 1, and the legs on the
heavy-tailed scenes."""
import glob
import hashlib
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
BENCH_PY = os.path.join(REPO, "bench.py")

HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
VALU_BOUND = ("k_draw", "k_draw_bwd", "k_draw_seg", "k_draw_bwd_seg")   # kernels whose roofline is VALU issue
XGMI_LINK_GBS = 153.0    # per link and direction, 7 links per GPU (prompt / SURVEY 8e)


def kernel_source_hash():
    """Fingerprint of the kernel sources: counters stored under profiles/ are only quoted when they were
    collected from exactly this code (there is no .git on the GPU box)."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(REPO, "easygaussiansplatting_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(REPO, "easygaussiansplatting_amd", "csrc", "*.h")) +
                    [os.path.join(REPO, "include", "egs_hip.h")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def algorithmic_bytes(kernel, N, P, T, HW, K, factored_views=0):
    """Algorithmic HBM bytes of ONE launch of `kernel` (each input read once, each output written once;
    atomics as read-modify-write).  ``factored_views`` > 0: the step keeps its SH
    gradient factored over that many views (the chain-rule kernel writes dL/dcolour + twc instead of the SH rows)."""
    nc = K // 3
    table = {
        # per-Gaussian stages with Jacobians: inputs + outputs + Jacobians
        "k_project": N * (12 + 8 + 12 + 4 + 24),
        "k_cov3d": N * (16 + 12 + 4 + 24 + 96 + 72),
        "k_cov2d": N * (24 + 12 + 4 + 12 + 72 + 36),
        "k_sh2color": N * (4 * K + 12 + 12 + 4 * nc + 36),
        "k_inv_cov2d": N * (12 + 4 + 12 + 8 + 36),
        "k_bin_count": N * (8 + 8 + 4 + 8 + 4 + 4),
        "k_bin_scan_partials": N * 8,                # depth-ordered packed rects (gathered by the depth sort's last scatter)
        "k_bin_scan_apply": N * (8 + 4),
        "k_pack_records": N * (36 + 8 + 48),
        "k_bin_emit": N * (4 + 4 + 8) + P * 8,
        "k_radix_hist": None, "k_radix_rowscan": None, "k_radix_scatter": None,  # size depends on the pass
        "k_tile_ranges": P * 4 + T * 8,
        "k_tile_order": T * 12,
        "k_tile_work": HW * 4 + T * 4,
        # draw, at the mandated op surface: 40 B per patch (u 8, cinv 12, alpha 4, color 12, gsid 4)
        # + ranges + 20 B per pixel out.  (The kernel gathers ONE packed 48-B record + 4-B list value instead.)
        "k_draw": 40 * P + 8 * T + 20 * HW,
        # drawB: the same gather + 9 fp32 atomics (RMW = 72 B) per patch + 20 B per pixel in
        "k_draw_bwd": 112 * P + 8 * T + 20 * HW,
        "k_chain_rule": N * (436 - 24 + 24 + 36 + 4 * (3 + 3 * nc + 3 + 4)),
        # fused path: parameters in (pw 12, rot 16, scale 12, sh 4K, alpha 4); out: depth 4, mask 1, the packed
        # 48-B record, the binning's compact record 16 -- and, in the training instance, dcolor/dpw 36 for the
        # backward kernel                                                          (= 341 N at K = 48)
        "k_preprocess_fwd": N * (44 + 4 * K + 4 + 1 + 48 + 16 + 36),
        # pw 12, rot 16, scale 12, depth 4, packed gradient record 48, dcolor/dpw 36 in (the SH rows are NOT read:
        # the forward kernel left dcolor/dpw); 59 gradient floats + du out        (= 372 N at K = 48)
        "k_preprocess_bwd": N * (40 + 4 + 48 + 36 + 4 * (3 + K + 1 + 3 + 4 + 2)),
        "k_unpack_grads": N * (48 + 36),
        # the SH rows of a step from its factored form: pw 12 + 12 per view in, 4K out
        "k_sh_grad_views": N * (12 + 12 * max(factored_views, 1) + 4 * K),
    }
    if factored_views > 0:   # 12 B of dL/dcolour instead of the 4K-byte row
        table["k_preprocess_bwd"] = N * (40 + 4 + 48 + 36 + 4 * (3 + 3 + 1 + 3 + 4 + 2))
    return table.get(kernel)


def parse_report(txt):
    out = {}
    for ln in txt.splitlines():
        parts = ln.split()
        if len(parts) == 3:
            out[parts[0]] = (int(parts[1]), float(parts[2]))
    return out


def cpu_baseline(scene, sample_n):
    """forward_cpu.py-equivalent (oracle policy A: vectorised stages + the
    per-Gaussian NumPy patch loop of gsplat/gausplat.py:185-245) on the first
    `sample_n` Gaussians of the iid scene, full resolution; cost is linear in N."""
    from oracle import gs_oracle as O
    sub = scene.subsample(slice(0, sample_n))
    cam = sub.cam
    t0 = time.perf_counter()
    P = O.POLICY_A
    us, pcs, depths = O.project(sub.pws, cam.Rcw, cam.tcw, cam.fx, cam.fy, cam.cx, cam.cy, P)
    cov3ds = O.compute_cov3d(sub.rots, sub.scales, depths, P)
    cov2ds = O.compute_cov2d(cov3ds, pcs, cam.Rcw, depths, cam.fx, cam.fy, cam.width, cam.height, P)
    colors = O.sh2color(sub.shs, sub.pws, cam.twc)
    cinv2ds, areas = O.inverse_cov2d(cov2ds, depths, P)
    O.splat_forward_cpu(cam.height, cam.width, us, cinv2ds, sub.alphas.astype(np.float64), depths, colors, areas)
    dt = time.perf_counter() - t0
    full = dt * scene.n / sample_n
    return {"value": round(cam.width * cam.height / full / 1e6, 5), "unit": "Mpix/s (forward only)",
            "cores": 1, "host_cores": os.cpu_count(), "kind": "port",
            "sample": ("all %d Gaussians at %dx%d, %.1f s measured, no extrapolation; "
                       % (scene.n, cam.width, cam.height, dt) if sample_n >= scene.n else
                       "first %d of %d iid Gaussians at %dx%d, %.1f s measured, x%.0f linear extrapolation; "
                       % (sample_n, scene.n, cam.width, cam.height, dt, scene.n / sample_n)) +
                      "single-threaded NumPy patch loop == reference forward_cpu.py"}


def relaunch_command(gpus, env, argv):
    """``python bench.py --gpus N`` (N > 1) started WITHOUT a launcher (no WORLD_SIZE / RANK in the environment):
    the argv of ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py <same flags>`` to exec instead -- one rank per GPU over RCCL, the very command the
    driver's contract names -- so that either way of starting the multi-GPU bench yields the one JSON line.
    None when no relaunch is due (N == 1, or already running under a launcher)."""
    if "WORLD_SIZE" in env or "RANK" in env or "LOCAL_RANK" in env:
        return None
    if gpus <= 1 and env.get("EGS_BENCH_FORCE_LAUNCHER") != "1":   # (the knob: exercise the re-exec path on a 1-GPU box)
        return None
    port = env.get("MASTER_PORT") or str(29500 + (os.getpid() % 2000))
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", port, BENCH_PY] + list(argv)


def _timed(fn, n, warm=2):
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def train_legs(sc, cam, dev, n_steps, full=False):
    """The whole training step on ``sc`` from camera ``cam`` (outside the timed region of the contract line):
      fwd_loss_bwd  render + fused HIP L1/SSIM loss + backward, no optimizer (GSFunction, activated parameters);
      train_step    raw parameters -> activations -> render -> loss -> backward -> Adam over 59 floats per Gaussian:
                    ``as_trainer_factored_sh`` is what ``Trainer.step`` enqueues for one view (deferred validation,
                    activations inside the kernels, the loss kernels hand dL/dimage to backward, the SH gradient stays
                    factored and FusedAdam forms the rows); ``fused_activations_fused_adam`` the same with dense SH rows;
                    ``full``: also torch activations and torch.optim.Adam (the reference's structure, gsmodel.py:198-210).
    -> (fwd_loss_bwd dict, train_step dict)"""
    import torch
    from easygaussiansplatting_amd import dist_views as DV, fused as fused_path
    from easygaussiansplatting_amd.function import GSFunction, GSRawFunction
    from easygaussiansplatting_amd.loss import gau_loss, gau_loss_with_grad
    from easygaussiansplatting_amd.optim import FusedAdam, adam_groups
    from easygaussiansplatting_amd.trainer import activate, raw_params_from_scene
    H, W = int(cam.height), int(cam.width)
    gt = torch.rand((3, H, W), device=dev)
    raw = raw_params_from_scene(sc, dev)
    act = [x.detach().clone().requires_grad_(True) for x in activate(raw)]
    us0 = torch.zeros((sc.n, 2), device=dev, requires_grad=True)

    def step_with_loss():
        for p in act:
            p.grad = None
        us0.grad = None
        with fused_path.deferred() as d:
            img, _ = GSFunction.apply(*act, us0, cam)
            gau_loss(img, gt).backward()
            d.commit()
    loss_ms = _timed(step_with_loss, n_steps, warm=4)
    del act
    opts = {"fused": FusedAdam(adam_groups(raw), eps=1e-15)}
    if full:
        opts["torch"] = torch.optim.Adam(adam_groups(raw), lr=0.0, eps=1e-15)

    def train_step(opt, fused_act=False):
        opt.zero_grad(set_to_none=True)
        us = torch.zeros((sc.n, 2), device=dev, requires_grad=True)
        if fused_act:   # activations inside the HIP kernels
            img, _ = GSRawFunction.apply(raw["pws"], raw["low_shs"], raw["high_shs"], raw["alphas_raw"],
                                         raw["scales_raw"], raw["rots_raw"], us, cam)
        else:           # the reference's structure: torch activations around GSFunction
            img, _ = GSFunction.apply(*activate(raw), us, cam)
        gau_loss(img, gt).backward()
        opt.step()
    fxt = DV.FactoredShGrad(1)
    us_keep = torch.zeros((sc.n, 2), device=dev, requires_grad=True)

    def train_step_factored(opt):
        opt.zero_grad(set_to_none=True)
        us_keep.grad = None
        with fused_path.deferred() as d, fxt.attach():
            img, _ = GSRawFunction.apply(raw["pws"], raw["low_shs"], raw["high_shs"], raw["alphas_raw"],
                                         raw["scales_raw"], raw["rots_raw"], us_keep, cam)
            _stats, dimg = gau_loss_with_grad(img.detach(), gt)
            img.backward(dimg)
            d.commit()
        rows, _w = fxt.take()
        opt.step(factored_sh=(rows, 1.0, raw["pws"], raw["low_shs"], raw["high_shs"]))
    tr = {"note": "1 view: activations + render + HIP loss + backward + Adam over 59 floats/Gaussian"}
    tr["train_step_ms_as_trainer_factored_sh"] = round(_timed(lambda: train_step_factored(opts["fused"]), n_steps, warm=4), 4)
    tr["train_step_ms_fused_activations_fused_adam"] = round(_timed(lambda: train_step(opts["fused"], True), n_steps), 4)
    tr["adam_only_ms_fused"] = round(_timed(opts["fused"].step, n_steps), 4)
    if full:
        for name, opt in opts.items():
            tr["train_step_ms_torch_activations_%s_adam" % name] = round(_timed(lambda: train_step(opt), n_steps), 4)
        tr["adam_only_ms_torch"] = round(_timed(opts["torch"].step, n_steps), 4)
    del raw, opts, us_keep, gt
    torch.cuda.empty_cache()
    return ({"ms": round(loss_ms, 4), "note": "render + fused HIP L1/SSIM loss + backward (no optimizer), 1 view"}, tr)


def epoch_pattern_leg(dev, epochs_with_history=2):
    """The access pattern of the reference's training loop (train.py:43-77) instead of one repeated camera: ONE view per
    optimizer step, the 8 ring cameras in a freshly shuffled order every epoch (train.py:44-49), a densification in the
    middle (N changes: every per-camera walk history, dispatch order and hint slot of the old size is dropped,
    gsmodel.py:214-317) and a ``reset_alpha`` after it (gsmodel.py:320-324: nothing saturates any more, tile walks
    grow from hundreds to thousands of entries).  The model is ``scene.skewed_scene`` (the shape of a trained scene:
    BASELINE configs[4]'s data is on no box), perturbed so that the loss has something to do; every step is
    ``Trainer.step`` (render + loss + backward + FusedAdam, deferred validation).  Per phase the mean GPU time of a step
    (HIP events around each step; the host runs ahead), first-sight epochs apart from epochs with history."""
    import torch
    from easygaussiansplatting_amd import scene as S
    from easygaussiansplatting_amd.function import Camera, render
    from easygaussiansplatting_amd.trainer import Trainer
    sc = S.skewed_scene()
    W, H = sc.cam.width, sc.cam.height
    cams = [Camera.from_scene(c, dev) for c in S.ring_cameras(sc.cam, 8)]
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
    with torch.no_grad():
        P = [t(sc.pws), t(sc.shs), t(sc.alphas), t(sc.scales), t(sc.rots)]
        gts = [render(*P, c)[0].clone() for c in cams]
        del P
    start = S.skewed_scene()
    start.pws[:] = start.pws + 0.004 * S.normal(11, 1, start.pws.shape).astype(np.float32)
    start.shs[:, :3] += 0.3 * S.normal(11, 2, (start.n, 3)).astype(np.float32)
    tr = Trainer(start, cams, gts, max_steps=3000, scene_size=8.0, seed=1)
    rng = np.random.default_rng(0)
    # One-off per PROCESS, like the first step of all: the first backward pass at an N that is no multiple of four zeroes
    # the <= 3 alignment words between the gradient slices with ``index_fill_`` (fused.backward), and PyTorch loads the
    # code object of that kernel on first use -- 45-55 ms of host time inside the first step after the first
    # densification (profiles/r6_densify_first_step.txt: `fused.backward 45.6 ms`, kernels of that step 1.8 ms; the
    # second densification's first step: 1.9 ms).  Loaded here, outside the timed steps.
    torch.zeros(8, device=dev).index_fill_(0, torch.tensor([1], device=dev), 0.0)
    out = {"what": "Trainer.step, ONE view per step, 8 ring cameras reshuffled per epoch; ms = mean GPU time per step "
                   "(HIP events); scene.skewed_scene perturbed (pws +- 0.004, SH degree 0 +- 0.3); PyTorch's "
                   "index_fill_ kernel (first use after the first densification: a 45-ms one-off module load per "
                   "process) is loaded before the timed steps", "phases": []}

    def epoch(label):
        evs = []
        for v in rng.permutation(8):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            tr.step([int(v)], sync=False)
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in evs]
        out["phases"].append({"phase": label, "gaussians": int(tr.params["pws"].shape[0]), "steps": len(ms),
                              "ms_per_step": round(float(np.mean(ms)), 4), "max_ms": round(float(np.max(ms)), 4),
                              "redone_steps_so_far": tr.redone_steps})
    epoch("first sight of every camera")
    for _ in range(epochs_with_history):
        epoch("with history")
    rep = tr.densify()
    out["densify"] = {k: (int(v) if isinstance(v, (int, np.integer)) else v) for k, v in (rep or {}).items()} \
        if isinstance(rep, dict) else str(rep)
    epoch("after densify (N changed): first sight again")
    epoch("after densify: with history")
    tr.reset_alpha()
    epoch("after reset_alpha: first epoch")
    epoch("after reset_alpha: with history")
    first = [p["ms_per_step"] for p in out["phases"] if "first" in p["phase"]]
    hist = [p["ms_per_step"] for p in out["phases"] if "first" not in p["phase"]]
    out["first_sight_ms_per_step"] = round(float(np.mean(first)), 4)
    out["with_history_ms_per_step"] = round(float(np.mean(hist)), 4)
    del tr, gts
    torch.cuda.empty_cache()
    return out


def uhd_leg(dev, lib, n=1_000_000, steps=10):
    """3840 x 2160 once (the reference is resolution-agnostic: kernel.cu:152, grid from gausplat.cu:94): T = 32 400 tiles
    (15 tile-key bits: two 8-bit passes of the tile sort; k_tile_order's tail registers), the bench scene seen through a
    camera of twice the focal length.  forward + backward, fused path, deferred validation."""
    import torch
    from easygaussiansplatting_amd import fused as fused_path, scene as S
    from easygaussiansplatting_amd.function import Camera, GSFunction
    W, H = 3840, 2160
    sc = S.big_scene(n, W, H, 48)
    sc.cam = S.Camera(W, H, 2400.0, 2400.0, W / 2.0, H / 2.0, sc.cam.Rcw, sc.cam.tcw)
    cam = Camera.from_scene(sc.cam, dev)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
    P = [t(sc.pws), t(sc.shs), t(sc.alphas).reshape(-1, 1).clone(), t(sc.scales), t(sc.rots)]
    for p in P:
        p.requires_grad_(True)
    us0 = torch.zeros((sc.n, 2), device=dev, requires_grad=True)
    dl = torch.from_numpy(S.normal(1, 78, (3, H, W)).astype(np.float32)).to(dev) / (3 * H * W)

    def step():
        for p in P:
            p.grad = None
        us0.grad = None
        with fused_path.deferred() as d:
            img, _ = GSFunction.apply(*P, us0, cam)
            img.backward(dl)
            if d.commit():
                img, _ = GSFunction.apply(*P, us0, cam)
                img.backward(dl)
    ms = _timed(step, steps, warm=6)
    with torch.no_grad():
        _, _, st = fused_path.forward(*[p.detach() for p in P], cam)
        lens = (st.ranges[:, 1] - st.ranges[:, 0]).to(torch.int64)
        out = {"width": W, "height": H, "gaussians": sc.n, "tiles": int(lens.numel()), "patches_drawn": int(st.patch_count()),
               "max_list_len": int(lens.max().item()), "ms_per_step": round(ms, 4),
               "Mpix/s": round(W * H / (ms * 1e-3) / 1e6, 2)}
    del P, us0, dl
    torch.cuda.empty_cache()
    return out


def scene_leg(name, sc, dev, lib, steps, iid_ref=None, train=False):
    """One extra leg of the default run, outside the timed region: the headline step (GSFunction fused, forward +
    backward, deferred validation) on ANOTHER scene -- list statistics, ms per step, the per-kernel table of the two
    draw kernels, and their time against the iid scene's scaled by the pixel-Gaussian pairs (every number
    of four rounds came from one iid distribution with lists <= 830).  ``iid_ref``: {"pairs", "k_draw_us",
    "k_draw_bwd_us"} of the headline scene."""
    import ctypes
    import torch
    from easygaussiansplatting_amd import fused as fused_path, scene as S
    from easygaussiansplatting_amd.function import Camera, GSFunction, RenderOptions
    cam = Camera.from_scene(sc.cam, dev)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
    P = dict(pws=t(sc.pws), shs=t(sc.shs), alphas=t(sc.alphas).reshape(-1, 1).clone(), scales=t(sc.scales),
             rots=t(sc.rots))
    for p in P.values():
        p.requires_grad_(True)
    us0 = torch.zeros((sc.n, 2), device=dev, requires_grad=True)
    W, H = sc.cam.width, sc.cam.height
    dl = torch.from_numpy(S.normal(1, 77, (3, H, W)).astype(np.float32)).to(dev) / (3 * H * W)

    def once(opts=None):
        for p in P.values():
            p.grad = None
        us0.grad = None
        args = (P["pws"], P["shs"], P["alphas"], P["scales"], P["rots"], us0, cam) + ((opts,) if opts is not None else ())
        img, _ = GSFunction.apply(*args)
        img.backward(dl)

    def step():
        with fused_path.deferred() as d:
            once()
            if d.commit():
                once()
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    lib.egs_prof_set_filter(None); lib.egs_prof_reset(); lib.egs_prof_enable(1)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    lib.egs_prof_enable(0)
    need = lib.egs_prof_report(None, 0)
    buf = ctypes.create_string_buffer(need + 16)
    lib.egs_prof_report(buf, need + 16)
    rep = parse_report(buf.value.decode())
    lib.egs_prof_reset()
    kern = {k: round(tot / c * 1e3, 2) for k, (c, tot) in sorted(rep.items(), key=lambda kv: -kv[1][1])}
    with torch.no_grad():
        d = {k: v.detach() for k, v in P.items()}
        _, _, st = fused_path.forward(d["pws"], d["shs"], d["alphas"], d["scales"], d["rots"], cam)
        lens = (st.ranges[:, 1] - st.ranges[:, 0]).to(torch.int64)
        walked = st.contrib.to(torch.int64)
        out = {"scene": name, "gaussians": sc.n, "patches_drawn": int(st.patch_count()),
               "max_list_len": int(lens.max().item()), "median_list_len": int(lens.median().item()),
               "pixel_gaussian_pairs": int(lens.sum().item()) * 256,
               # what the blend loops really walk: per pixel the index of its last contributor (early termination)
               "walked_pairs": int(walked.sum().item()), "max_walked": int(walked.max().item()),
               "ms_per_step": round(ms, 4), "Mpix/s": round(W * H / (ms * 1e-3) / 1e6, 2), "kernels_avg_us": kern}
    # the draw stage of either path: the unsplit kernel, or the segment kernels + the planning launch in front of them
    # and the one-word report behind them (the backward launch runs over the forward pass's work items: no plan of its own)
    fwd = sum(v for k, v in kern.items() if (k.startswith("k_draw") and "bwd" not in k) or k.startswith("k_seg_"))
    bwd = sum(v for k, v in kern.items() if k.startswith("k_draw_bwd"))
    out["draw_fwd_us"], out["draw_bwd_us"] = round(fwd, 1), round(bwd, 1)
    out["segment_path"] = "k_draw_seg" in kern
    # the seven-op drop-in surface on the same scene: with this package's records handle, and the plain public
    # splat / splatB pair of an unmodified reference caller (which rebuilds the segment states in splatB)
    for key, o in (("ops_ms_per_step", RenderOptions(mode="ops")),
                   ("ops_public_pair_ms_per_step", RenderOptions(mode="ops", ops_use_records=False))):
        for _ in range(4):
            once(o)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            once(o)
        torch.cuda.synchronize()
        out[key] = round((time.perf_counter() - t0) / 8 * 1e3, 4)
    if out["segment_path"]:
        # the public pair's three settings (gsplatcu.set_pair_states): the default above keeps the forward's segment
        # states and compares the values splatB is handed with a snapshot on the device; True matches by identity and
        # version only; False keeps nothing (splatB rebuilds the states from contrib)
        from easygaussiansplatting_amd import gsplatcu as _gsc
        out["ops_public_pair_states"] = dict(_gsc.last_splatB_info())
        o = RenderOptions(mode="ops", ops_use_records=False)
        for key, mode in (("ops_public_pair_identity_matched_ms_per_step", True),
                          ("ops_public_pair_rebuilding_ms_per_step", False)):
            prev = _gsc.set_pair_states(mode)
            try:
                for _ in range(4):
                    once(o)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(8):
                    once(o)
                torch.cuda.synchronize()
                out[key] = round((time.perf_counter() - t0) / 8 * 1e3, 4)
            finally:
                _gsc.set_pair_states(prev)
    if iid_ref:
        r = out["pixel_gaussian_pairs"] / iid_ref["pairs"]
        out["pairs_ratio_to_iid"] = round(r, 3)
        for k, v in (("k_draw", fwd), ("k_draw_bwd", bwd)):
            if iid_ref.get(k + "_us"):      # VERDICT r4 #1's yardstick: the iid scene's kernel time scaled by the pairs
                out[k + "_over_pairs_scaled_iid"] = round(v / (r * iid_ref[k + "_us"]), 3)
    del P, us0, dl
    torch.cuda.empty_cache()
    if train:     # the whole training step on this scene (render + loss + backward + Adam)
        out["fwd_loss_bwd"], out["train_step"] = train_legs(sc, cam, dev, 8)
    return out

