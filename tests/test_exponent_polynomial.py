"""CPU check of the arithmetic k_draw uses for the exponent of alpha' (csrc/egs_draw.hip): the quadratic
form evaluated as a polynomial about the TILE CENTRE with per-lane constant monomials,
    e = c0 + c1 X + c2 Y + qxx XX + qxy XY + qyy YY,   c0 = log2(alpha) + E(D), (c1, c2) = grad E(D),
against the direct form E(u - pixel) (what the reference computes, common.cuh:85-88) -- both emulated in fp32
with the kernel's operation order and compared with fp64.  The polynomial has to stay two orders of
magnitude inside the image tolerance; the GPU parity tests then pin the kernel itself against the oracle."""
import numpy as np

from easygaussiansplatting_amd import scene as S
from oracle import gs_oracle as O

f = np.float32
NHL2E = f(-0.72134752044)          # -0.5 log2(e): the record's pre-scaled conic (egs_gaussian_math.h)


def _records(sc):
    cam, P = sc.cam, O.POLICY_G
    us, pcs, depths = O.project(sc.pws, cam.Rcw, cam.tcw, cam.fx, cam.fy, cam.cx, cam.cy, P, False, f)
    cov3 = O.compute_cov3d(sc.rots, sc.scales, depths, P, False, f)
    cov2 = O.compute_cov2d(cov3, pcs, cam.Rcw, depths, cam.fx, cam.fy, cam.width, cam.height, P, False, f)
    cinv, areas = O.inverse_cov2d(cov2, depths, P, False, f)
    rects, counts = O.get_rects(us, areas, depths, cam.width, cam.height, P)[:2]
    return us, cinv, sc.alphas.reshape(-1).astype(f), np.asarray(rects).reshape(-1, 4).astype(np.int64)


def _patches(rects, limit, seed=0):
    x0, y0, x1, y1 = rects.T
    w = x1 - x0
    cnt = w * (y1 - y0)
    gid = np.repeat(np.arange(len(cnt)), cnt)
    off = np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    tx = x0[gid] + off % w[gid]
    ty = y0[gid] + off // w[gid]
    if len(gid) > limit:
        sel = np.random.default_rng(seed).choice(len(gid), limit, replace=False)
        gid, tx, ty = gid[sel], tx[sel], ty[sel]
    return gid, tx, ty


def _errors(sc, limit=60000):
    us, cinv, alpha, rects = _records(sc)
    gid, tx, ty = _patches(rects, limit)
    q = lambda a: a[:, None, None]
    qxx = q((NHL2E * cinv[gid, 0]).astype(f)); qxy = q((f(2) * NHL2E * cinv[gid, 1]).astype(f))
    qyy = q((NHL2E * cinv[gid, 2]).astype(f))
    ux, uy = q(us[gid, 0].astype(f)), q(us[gid, 1].astype(f))
    lskip = f(np.log2(0.002))
    la = q((lskip - np.log2(f(0.002) / alpha[gid]).astype(f)).astype(f))     # log2(alpha) as the kernel derives it
    l16 = np.arange(16, dtype=f)
    px = q((tx * 16).astype(f)) + l16[None, None, :]
    py = q((ty * 16).astype(f)) + l16[None, :, None]
    d = np.float64
    dx64, dy64 = px.astype(d) - ux.astype(d), py.astype(d) - uy.astype(d)
    e64 = qxx.astype(d) * dx64 * dx64 + qxy.astype(d) * dx64 * dy64 + qyy.astype(d) * dy64 * dy64 + la.astype(d)
    # direct: cxx = (qxx dx) dx + la; cyy = qyy dy dy; e = cxx + cyy + (qxy dx) dy
    dx, dy = (ux - px).astype(f), (uy - py).astype(f)
    e_dir = (((qxx * dx) * dx + la) + (qyy * dy) * dy + (qxy * dx) * dy).astype(f)
    # polynomial about the tile centre: coefficients in fp32 (the staging lane), five FMAs per pixel
    cx0, cy0 = q((tx * 16).astype(f) + f(7.5)), q((ty * 16).astype(f) + f(7.5))
    Dx, Dy = (cx0 - ux).astype(f), (cy0 - uy).astype(f)
    c0 = (la + (qxx * Dx * Dx + qxy * Dx * Dy + qyy * Dy * Dy)).astype(f)
    c1 = (f(2) * qxx * Dx + qxy * Dy).astype(f); c2 = (f(2) * qyy * Dy + qxy * Dx).astype(f)
    X = (l16 - f(7.5))[None, None, :]; Y = (l16 - f(7.5))[None, :, None]
    e = (c2 * Y + c0).astype(f); e = (c1 * X + e).astype(f); e = (qyy * (Y * Y).astype(f) + e).astype(f)
    e = (qxy * (X * Y).astype(f) + e).astype(f); e_pol = (qxx * (X * X).astype(f) + e).astype(f)
    hit = e64 >= np.log2(0.002)                                  # the pixels that blend (kernel.cu:246)
    flips = lambda a: int(((a >= lskip) != hit).sum())
    return np.abs(e_dir - e64)[hit], np.abs(e_pol - e64)[hit], flips(e_dir), flips(e_pol), int(hit.sum())


def test_polynomial_exponent_stays_far_inside_the_tolerance():
    err_dir, err_pol, fl_dir, fl_pol, nhit = _errors(S.small_scene(10000, 256, 256, 3, seed=0))
    assert nhit > 400_000
    # errors in the log2 domain: 1e-4 there is 7e-5 relative in alpha' (the image tolerance is 1e-4 absolute).
    # Sub-pixel Gaussians (sigma = 0.55 px after the +0.3 dilation, qxx = -2.4) are the polynomial's worst case:
    # its terms reach 2 |qxx| 7.5^2 = 270 where the differences stay below 20, so it is a few times less
    # accurate THERE (numpy has no FMA: the kernel's single-rounding FMAs halve these numbers) -- and still two
    # orders of magnitude inside the tolerance.
    assert err_dir.mean() < 1e-6
    assert err_pol.mean() < 3e-6 and np.quantile(err_pol, 0.999) < 3e-5 and err_pol.max() < 2e-4, \
        (err_pol.mean(), np.quantile(err_pol, 0.999), err_pol.max())
    assert fl_pol <= fl_dir + max(4, nhit // 200_000)           # skip-threshold flips stay as rare


def test_polynomial_exponent_on_needles():
    """Condition numbers ~2000 (long thin Gaussians at every angle): neither form is exact, the polynomial
    is not the worse one."""
    sc = S.small_scene(4000, 320, 208, 3, seed=33)
    sc.scales[:, 0] = 0.4
    sc.scales[:, 1:] = 0.004
    err_dir, err_pol, fl_dir, fl_pol, nhit = _errors(sc)
    assert nhit > 200_000
    assert err_pol.mean() <= 1.5 * err_dir.mean() and np.quantile(err_pol, 0.999) <= 1.5 * np.quantile(err_dir, 0.999)
    assert err_pol.max() < 5e-3
