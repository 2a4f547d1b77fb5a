#!/usr/bin/env python3
"""Generate the golden fixtures G1-G6 by IMPORTING the
reference's own Python in the build container.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden.py [--g6]

Only *data* (inputs + the reference's outputs) is written, as small ``.npz``
files next to this script.  ``/root/reference`` is read-only and does not exist
on the GPU box; nothing in the test-suite imports it -- the tests read the
fixtures.  The reference functions called are named in each fixture's
``__doc__`` entry.

G5 additionally uses the repo's own oracle for the *integer* tile lists
(bit-exact spec, tested separately); every float in the fixtures comes from
the reference's code.
"""
import argparse
import os
import sys
import types

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

# the reference imports plyfile (absent) and gsplatcu (the CUDA extension) at
# module import time (gau_io.py:2, utils.py:2); neither is used by the
# functions called here.
ply = types.ModuleType("plyfile"); ply.PlyData = object
sys.modules["plyfile"] = ply
sys.modules["gsplatcu"] = types.ModuleType("gsplatcu")
sys.path.insert(0, REF)
sys.path.insert(1, REPO)

import numpy as np  # noqa: E402

import gsplat.gausplat as ref_a  # noqa: E402  (oracle A)
import backward_cpu as ref_b  # noqa: E402      (oracle B)

from easygaussiansplatting_amd import scene as S  # noqa: E402
from oracle import gs_oracle as O  # noqa: E402
from tests.golden.make_golden_scene import stage_scene  # noqa: E402


from tests.golden import _recipe  # noqa: E402
from tests.golden._recipe import save  # noqa: E402

_recipe.assert_reference(ref_a, ref_b)      # NOT this repository's same-named modules


def g1():
    """backward_cpu.py stage functions with calc_J=True, float64."""
    sc = stage_scene()
    cam = sc.cam
    n = sc.n
    twc = np.linalg.inv(cam.Rcw) @ (-cam.tcw)
    pws = sc.pws.astype(np.float64); rots = sc.rots.astype(np.float64)
    scales = sc.scales.astype(np.float64); shs = sc.shs.astype(np.float64)
    out = {k: [] for k in ("pcs", "dpc_dpws", "us", "du_dpcs", "cov3ds", "dcov3d_drots", "dcov3d_dscales",
                           "cov2ds", "dcov2d_dcov3ds", "dcov2d_dpcs", "colors", "dcolor_dshs", "dcolor_dpws",
                           "cinv2ds", "dcinv2d_dcov2ds")}
    for i in range(n):
        pc, dpc = ref_b.transform(pws[i], cam.Rcw, cam.tcw, True)
        u, du = ref_b.project(pc, cam.fx, cam.fy, cam.cx, cam.cy, True)
        c3, dq, ds = ref_b.compute_cov_3d(rots[i], scales[i], True)
        c2, d3, dpcj = ref_b.compute_cov_2d(c3, pc, cam.Rcw, cam.fx, cam.fy, True)
        col, dsh, dpw = ref_b.sh2color(shs[i], pws[i], twc, True)
        ci, dci = ref_b.calc_cinv2d(c2, True)
        for k, v in zip(out.keys(), (pc, dpc, u, du, c3, dq, ds, c2, d3, dpcj, col, dsh, dpw, ci, dci)):
            out[k].append(np.array(v))
    out = {k: np.stack(v) for k, v in out.items()}
    # lower SH degrees on the same Gaussians
    for K in (3, 12, 27):
        cols, dshs, dpws = [], [], []
        for i in range(n):
            col, dsh, dpw = ref_b.sh2color(shs[i, :K], pws[i], twc, True)
            cols.append(col); dshs.append(dsh); dpws.append(dpw)
        out["colors_K%d" % K] = np.stack(cols)
        out["dcolor_dshs_K%d" % K] = np.stack(dshs)
        out["dcolor_dpws_K%d" % K] = np.stack(dpws)
    # known-answer inputs of the reference's scratch tests
    # (test/test_cov3d.py:112-113, test/test_cov2d.py:104-110)
    q = np.array([0.606, -0.002, -0.755, 0.252]); s = np.array([1.2, 3.2, 0.5])
    ka3, ka3q, ka3s = ref_b.compute_cov_3d(q, s, True)
    kR = np.array([[-0.267058, -0.302404, -0.916068], [0.308444, 0.872984, -0.378096],
                   [0.914052, -0.382944, -0.140058]])
    kpc = np.array([1.0, 2.0, 3.0])
    ka2, ka2d3, ka2dpc = ref_b.compute_cov_2d(ka3, kpc, kR, 200.0, 100.0, True)
    save("g1_stages_b.npz", "reference backward_cpu.py: transform/project/compute_cov_3d/compute_cov_2d/"
         "sh2color/calc_cinv2d with calc_J=True on 256 seeded Gaussians (float64)",
         pws=sc.pws, rots=sc.rots, scales=sc.scales, alphas=sc.alphas, shs=sc.shs,
         Rcw=cam.Rcw, tcw=cam.tcw, twc=twc, intr=np.array([cam.fx, cam.fy, cam.cx, cam.cy, cam.width, cam.height]),
         ka_q=q, ka_s=s, ka_cov3d=ka3, ka_dcov3d_dq=ka3q, ka_dcov3d_ds=ka3s, ka_Rcw=kR, ka_pc=kpc,
         ka_cov2d=ka2, ka_dcov2d_dcov3d=ka2d3, ka_dcov2d_dpc=ka2dpc, **out)


def g2():
    """gsplat/gausplat.py stage functions (oracle A) on the same Gaussians."""
    sc = stage_scene()
    cam = sc.cam
    twc = np.linalg.inv(cam.Rcw) @ (-cam.tcw)
    us, pcs = ref_a.project(sc.pws, cam.Rcw, cam.tcw, cam.fx, cam.fy, cam.cx, cam.cy)
    cov3ds = ref_a.compute_cov_3d(sc.scales, sc.rots)
    with np.errstate(all="ignore"):
        cov2ds = ref_a.compute_cov_2d(pcs, cam.fx, cam.fy, cam.width, cam.height, cov3ds, cam.Rcw)
        colors = ref_a.sh2color(sc.shs, sc.pws, twc)
        cinv2ds, areas = ref_a.inverse_cov2d(cov2ds)
    save("g2_stages_a.npz", "reference gsplat/gausplat.py: project/compute_cov_3d/compute_cov_2d/sh2color/"
         "inverse_cov2d on the G1 Gaussians (float32 inputs as forward_cpu.py feeds them)",
         us=us, pcs=pcs, cov3ds=cov3ds, cov2ds=cov2ds, colors=colors, cinv2ds=cinv2ds, areas=areas)


def g3():
    """Everything backward_gpu.py:52-152 compares, on get_example_gs()."""
    sc = S.example_gs()
    cam = sc.cam
    gs_num = 4
    np.random.seed(0)
    rest = np.random.rand(gs_num, 45)           # the reference is unseeded here (backward_cpu.py:507)
    shs = np.concatenate((sc.shs, rest), axis=1).astype(np.float64)
    pws = sc.pws.astype(np.float64); alphas = sc.alphas.astype(np.float64)
    rots = sc.rots.astype(np.float64); scales = sc.scales.astype(np.float64)
    Rcw, tcw = cam.Rcw, cam.tcw
    twc = np.linalg.inv(Rcw) @ (-tcw)
    W, H = cam.width, cam.height
    image_gt = np.zeros([H, W, 3])
    names = ("pcs", "dpc_dpws", "us", "du_dpcs", "cov3ds", "dcov3d_drots", "dcov3d_dscales", "cov2ds",
             "dcov2d_dcov3ds", "dcov2d_dpcs", "colors", "dcolor_dshs", "dcolor_dpws", "cinv2ds", "dcinv2d_dcov2ds")
    out = {k: [] for k in names}
    for i in range(gs_num):
        pc, dpc = ref_b.transform(pws[i], Rcw, tcw, True)
        u, du = ref_b.project(pc, cam.fx, cam.fy, cam.cx, cam.cy, True)
        c3, dq, ds = ref_b.compute_cov_3d(rots[i], scales[i], True)
        c2, d3, dpcj = ref_b.compute_cov_2d(c3, pc, Rcw, cam.fx, cam.fy, True)
        col, dsh, dpw = ref_b.sh2color(shs[i], pws[i], twc, True)
        ci, dci = ref_b.calc_cinv2d(c2, True)
        for k, v in zip(names, (pc, dpc, u, du, c3, dq, ds, c2, d3, dpcj, col, dsh, dpw, ci, dci)):
            out[k].append(np.array(v))
    out = {k: np.stack(v) for k, v in out.items()}
    image = ref_b.get_image(alphas, out["cinv2ds"], out["colors"], out["us"], H, W)
    loss, dl_da, dl_dci, dl_dc, dl_du = ref_b.calc_loss(alphas, out["cinv2ds"], out["colors"], out["us"],
                                                        image_gt, True)
    _, dloss_dgammas = ref_b.get_loss(image, image_gt)
    ref_b.sh_dim = 48                            # module global read by backward() (backward_cpu.py:456)
    loss2, drots, dscales, dshs, dalphas, dpws = ref_b.backward(rots, scales, shs, alphas, pws, Rcw, tcw,
                                                                cam.fx, cam.fy, cam.cx, cam.cy, image_gt, True)
    save("g3_example_backward.npz", "reference backward_cpu.py on get_example_gs(), 32x16, fx=fy=16, "
         "np.random.seed(0) SH rest: all arrays backward_gpu.py:83-152 checks + backward() parameter grads",
         shs=shs, pws=pws, alphas=alphas, rots=rots, scales=scales, Rcw=Rcw, tcw=tcw, twc=twc,
         image=image, loss=loss, dloss_dgammas=dloss_dgammas,
         dloss_dalphas=dl_da.reshape(gs_num, 1, 1), dloss_dcinv2ds=dl_dci.reshape(gs_num, 1, 3),
         dloss_dcolors=dl_dc.reshape(gs_num, 1, 3), dloss_dus=dl_du.reshape(gs_num, 1, 2),
         dloss_drots=drots, dloss_dscales=dscales, dloss_dshs=dshs, dloss_dalphas_final=dalphas,
         dloss_dpws=dpws, **out)


def run_forward_cpu(sc):
    """forward_cpu.py:43-60 verbatim call sequence on a Scene."""
    cam = sc.cam
    gs = sc.as_records()
    twc = np.linalg.inv(cam.Rcw) @ (-cam.tcw)
    pws = gs['pw']
    us, pcs = ref_a.project(pws, cam.Rcw, cam.tcw, cam.fx, cam.fy, cam.cx, cam.cy)
    depths = pcs[:, 2]
    cov3ds = ref_a.compute_cov_3d(gs['scale'], gs['rot'])
    cov2ds = ref_a.compute_cov_2d(pcs, cam.fx, cam.fy, cam.width, cam.height, cov3ds, cam.Rcw)
    colors = ref_a.sh2color(gs['sh'], pws, twc)
    cinv2ds, areas = ref_a.inverse_cov2d(cov2ds)
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        image = ref_a.splat(cam.height, cam.width, us, cinv2ds, gs['alpha'], depths, colors, areas)
    return dict(us=us, pcs=pcs, depths=depths, cov3ds=cov3ds, cov2ds=cov2ds, colors=colors,
                cinv2ds=cinv2ds, areas=areas, image=image)


def g4():
    sc = S.small_scene()
    r = run_forward_cpu(sc)
    k = 64
    save("g4_forward_cpu_10k.npz", "reference forward_cpu.py pipeline (gsplat/gausplat.py) on "
         "scene.small_scene(): N=10k, 256x256, SH degree 0 (BASELINE configs[0]); image float32 [H,W,3], "
         "stage arrays of the first 64 Gaussians float64",
         image=r["image"].astype(np.float32), us=r["us"][:k], pcs=r["pcs"][:k], cov3ds=r["cov3ds"][:k],
         cov2ds=r["cov2ds"][:k], colors=r["colors"][:k], cinv2ds=r["cinv2ds"][:k], areas=r["areas"][:k],
         us_all=r["us"].astype(np.float32))


def g5():
    """Oracle-B blend fwd+bwd on a multi-tile scene with per-pixel lists
    restricted to the gsplatcu tile lists."""
    W, H = 48, 32
    n = 160
    seed = 21
    u01 = lambda s, shp: S.uniform01(seed, s, shp)
    us = np.stack([-4 + (W + 8) * u01(1, (n,)), -4 + (H + 8) * u01(2, (n,))], 1)
    sx = 0.8 + 5.0 * u01(3, (n,)); sy = 0.8 + 5.0 * u01(4, (n,)); rho = -0.7 + 1.4 * u01(5, (n,))
    cov2ds = np.stack([sx * sx, rho * sx * sy, sy * sy], 1)
    alphas = 0.05 + 0.94 * u01(6, (n,))
    colors = 0.05 + 0.9 * u01(7, (n, 3))
    # an opaque cluster among the nearest Gaussians around (30,14) so that tau
    # drops below 1e-4 (early stop, kernel.cu:256) with more Gaussians behind it
    k = np.arange(0, 40, 2)
    us[k] = np.stack([30 + 6 * (u01(10, (k.size,)) - 0.5), 14 + 6 * (u01(11, (k.size,)) - 0.5)], 1)
    sx[k] = 5 + 3 * u01(12, (k.size,)); sy[k] = 5 + 3 * u01(13, (k.size,))
    cov2ds = np.stack([sx * sx, rho * sx * sy, sy * sy], 1)
    alphas[k] = 0.8 + 0.19 * u01(14, (k.size,))
    depths = 0.5 + np.arange(n) * 0.05                  # pre-sorted, strictly increasing by 50 mm
    f32 = np.float32
    us32, cov32, al32, col32, dep32 = (x.astype(f32) for x in (us, cov2ds, alphas, colors, depths))
    us, cov2ds, alphas, colors = (x.astype(np.float64) for x in (us32, cov32, al32, col32))
    cinv2ds = np.stack([ref_b.calc_cinv2d(c) for c in cov2ds])
    # integer tile lists: the repo's own restatement of getRects/createKeys/sort (bit-exact spec)
    cinv_o, areas = O.inverse_cov2d(cov32, dep32.copy(), O.POLICY_G)
    ranges, gsid, rects, counts = O.bin_tiles(us32, areas.copy(), dep32.copy(), W, H, O.POLICY_G)
    gx, gy = O.tile_grid(W, H)
    rng_gt = S.normal(seed, 9, (3, H, W)) / (H * W)
    image = np.zeros((H, W, 3)); contrib = np.zeros((H, W), np.int32); tau_img = np.zeros((H, W))
    dalphas = np.zeros(n); dcinv = np.zeros((n, 3)); dcolors = np.zeros((n, 3)); dus = np.zeros((n, 2))
    for py in range(H):
        for px in range(W):
            t = (py // 16) * gx + (px // 16)
            lst = gsid[ranges[t, 0]:ranges[t, 1]]
            if lst.size == 0:
                continue
            x = np.array([px, py])
            gamma, dg_da, dg_dci, dg_dc, dg_du, cont = ref_b.calc_gamma(
                alphas[lst], cinv2ds[lst], colors[lst], us[lst], x, True)
            image[py, px] = gamma
            contrib[py, px] = cont
            # final tau: replay the forward recurrence of calc_gamma (backward_cpu.py:241-250)
            tau = 1.0
            for a, ci, uu in zip(alphas[lst][:cont], cinv2ds[lst][:cont], us[lst][:cont]):
                ap = ref_b.calc_alpha_prime(a, ci, uu, x)
                if ap < 0.002:
                    continue
                tau *= (1 - ap)
            tau_img[py, px] = tau
            dl = rng_gt[:, py, px]
            for j in range(cont):        # backward_cpu.py:426-430
                g = lst[j]
                dalphas[g] += (dl @ dg_da[j]).item()
                dcinv[g] += dl @ dg_dci[j]
                dcolors[g] += dl @ dg_dc[j]
                dus[g] += dl @ dg_du[j]
    save("g5_raster_b_multitile.npz", "reference backward_cpu.py calc_gamma(calc_J=True) per pixel on a "
         "48x32 (3x2 tiles) scene of 160 2D Gaussians, per-pixel lists = gsplatcu tile lists; "
         "dL/dimage = seeded N(0,1)/HW",
         us=us32, cov2ds=cov32, cinv2ds=cinv2ds, alphas=al32, colors=col32, depths=dep32, areas=areas,
         ranges=ranges, gsid=gsid, rects=rects, counts=counts, dloss_dgammas=rng_gt,
         image=image, contrib=contrib, final_tau=tau_img,
         dloss_dalphas=dalphas, dloss_dcinv2ds=dcinv, dloss_dcolors=dcolors, dloss_dus=dus)


def g6():
    sc = S.big_scene()
    r = run_forward_cpu(sc)
    img = r["image"]                       # [H,W,3] f64
    H, W = sc.cam.height, sc.cam.width
    gx, gy = O.tile_grid(W, H)
    pad = np.zeros((gy * 16, gx * 16, 3)); pad[:H, :W] = img
    cnt = np.zeros((gy * 16, gx * 16)); cnt[:H, :W] = 1
    tm = pad.reshape(gy, 16, gx, 16, 3).sum((1, 3)) / cnt.reshape(gy, 16, gx, 16).sum((1, 3))[..., None]
    sel = (S.uniform01(3, 1, (64,)) * (gx * (gy - 1))).astype(np.int64)   # full tiles only
    tiles = np.stack([pad[(t // gx) * 16:(t // gx) * 16 + 16, (t % gx) * 16:(t % gx) * 16 + 16] for t in sel])
    save("g6_forward_cpu_1m_digest.npz", "reference forward_cpu.py pipeline on scene.big_scene(): N=1M, "
         "1920x1080, SH degree 3 (BASELINE configs[1]); digest = per-tile mean RGB [68,120,3], 64 full "
         "16x16 tiles, radii histogram",
         tile_mean=tm.astype(np.float32), tile_ids=sel, tiles=tiles.astype(np.float32),
         areas_hist=np.bincount(np.clip(r["areas"].reshape(-1), 0, 255), minlength=256),
         image_mean=np.array(img.mean()), image_absmax=np.array(np.abs(img).max()))


def g7():
    """gsplat/pytorch_ssim.py gau_loss + torch autograd gradient (CPU) on seeded image pairs."""
    import torch
    import gsplat.pytorch_ssim as ref_l
    _recipe.assert_reference(ref_l)
    out = {}
    for tag, (H, W) in (("a", (37, 53)), ("b", (16, 64)), ("c", (9, 7))):
        x = (0.5 + 0.35 * S.normal(31, 1, (3, H, W))).astype(np.float32)
        y = np.clip(x + 0.15 * S.normal(31, 2, (3, H, W)), 0, 1).astype(np.float32)
        y[:, : H // 3] = x[:, : H // 3]                     # identical region: sign(0) = 0 branch of |x-y|
        xt = torch.from_numpy(x).double().requires_grad_(True)
        yt = torch.from_numpy(y).double()
        loss = ref_l.gau_loss(xt, yt)
        loss.backward()
        out["x_" + tag] = x; out["y_" + tag] = y
        out["loss_" + tag] = loss.detach().numpy(); out["grad_" + tag] = xt.grad.numpy()
        out["ssim_" + tag] = ref_l.ssim(xt.detach(), yt).numpy()
        xf = torch.from_numpy(x).requires_grad_(True)
        lf = ref_l.gau_loss(xf, torch.from_numpy(y)); lf.backward()
        out["loss32_" + tag] = lf.detach().numpy(); out["grad32_" + tag] = xf.grad.numpy()
    save("g7_gau_loss.npz", "reference gsplat/pytorch_ssim.py gau_loss(image, gt) and d loss/d image by torch "
         "autograd (float64 and float32, CPU) on three seeded [3,H,W] image pairs", **out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--g6", action="store_true", help="also run the 1M/1080p reference forward (~1 min)")
    ap.add_argument("--check", action="store_true",
                    help="regenerate into a temp dir and compare with the committed fixtures (exit 1 on a difference)")
    a = ap.parse_args()
    _recipe.begin(a.check)
    todo = [g1, g2, g3, g4, g5, g7] + ([g6] if a.g6 else [])
    for fn in todo:
        if a.only and fn.__name__ not in a.only.split(","):
            continue
        print("==", fn.__name__)
        fn()
    sys.exit(_recipe.finish())
