"""Why do four Gaussians of the 1 M bench scene carry a 0.5-0.8 % error in one channel of dL/dcolour against the oracle
fed numpy-float32 stages, unflagged by every threshold-flip margin?  For each of them: the device's 2D Gaussian against
numpy's float32 one (in ulps), and the pixels whose alpha' >= 0.002 decision differs between the two parameter sets."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from easygaussiansplatting_amd import gsplatcu as gsc, scene as S      # noqa: E402
from oracle import gs_oracle as O                                      # noqa: E402
from tests.test_gpu_parity import _oracle_2d                           # noqa: E402

rows = np.array([int(x) for x in sys.argv[1:]] or [84715, 654979, 657494, 938602])
sc = S.big_scene()
cam = sc.cam
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
host = lambda t: t.detach().cpu().numpy()
gsc.set_policy("gsplatcu")
sub = sc.subsample(rows)
pws, rots, scales, shs = map(dev, (sub.pws, sub.rots, sub.scales, sub.shs))
Rcw, tcw, twc = dev(cam.Rcw), dev(cam.tcw), dev(cam.twc)
us, pcs, depths = gsc.project(pws, Rcw, tcw, cam.fx, cam.fy, cam.cx, cam.cy, False)
cov3 = gsc.computeCov3D(rots, scales, depths, False)[0]
cov2 = gsc.computeCov2D(cov3, pcs, Rcw, depths, cam.fx, cam.fy, cam.width, cam.height, False)[0]
cinv, areas = gsc.inverseCov2D(cov2, depths, False)
d_us, d_ci, d_c2 = host(us).astype(np.float64), host(cinv).astype(np.float64), host(cov2).astype(np.float64)
o_us, o_ci, _, _, _ = _oracle_2d(sc, cam, rows, dtype=np.float32)
q_us, q_ci, _, _, _ = _oracle_2d(sc, cam, rows, dtype=np.float64)
for i, g in enumerate(rows):
    a = float(sc.alphas[g])
    ulp = lambda x: 2.0 ** (np.floor(np.log2(np.abs(x) + 1e-300)) - 23)
    print("gaussian %d alpha %.4f scales %s  u (device) %s" % (g, a, sc.scales[g], d_us[i]))
    print("   u: device - numpy32 = %s ulps, device - float64 = %s ulps" % ((d_us[i] - o_us[i]) / ulp(d_us[i]), (d_us[i] - q_us[i]) / ulp(d_us[i])))
    print("   cinv device %s\n   cinv numpy32 rel diff %s   float64 rel diff %s" % (d_ci[i], (d_ci[i] - o_ci[i]) / np.abs(d_ci[i]), (d_ci[i] - q_ci[i]) / np.abs(d_ci[i])))
    print("   cov2d device %s  det %.6g  cond ~ %.1f" % (d_c2[i], d_c2[i][0] * d_c2[i][2] - d_c2[i][1] ** 2,
                                                          (d_c2[i][0] + d_c2[i][2]) ** 2 / (d_c2[i][0] * d_c2[i][2] - d_c2[i][1] ** 2)))
    r = int(np.ceil(3 * np.sqrt(max(d_c2[i][0], d_c2[i][2])))) + 1
    xs = np.arange(int(d_us[i][0]) - r, int(d_us[i][0]) + r + 1, dtype=np.float64)
    ys = np.arange(int(d_us[i][1]) - r, int(d_us[i][1]) + r + 1, dtype=np.float64)
    py, px = np.meshgrid(ys, xs, indexing="ij")

    def ap(u, c):
        dx = u[0] - px; dy = u[1] - py
        m = np.maximum(c[0] * dx * dx + c[2] * dy * dy + 2 * c[1] * dx * dy, 0)
        return np.minimum(a * np.exp(-0.5 * m), 0.99)
    ad, an = ap(d_us[i], d_ci[i]), ap(o_us[i], o_ci[i])
    flip = (ad >= 0.002) != (an >= 0.002)
    rel = np.abs(an - 0.002) / 0.002
    k = np.argsort(rel.ravel())[:4]
    print("   pixels whose decision differs: %d; closest to the threshold (numpy32 params): rel distance %s, device-vs-numpy alpha' rel diff there %s"
          % (int(flip.sum()), rel.ravel()[k], (np.abs(ad - an) / an).ravel()[k]))

# ---- the fused path's culled lists and block masks for these Gaussians: is a contributing 8x8 block (or tile) missing?
from easygaussiansplatting_amd import fused                           # noqa: E402
from easygaussiansplatting_amd.function import Camera                  # noqa: E402
with torch.no_grad():
    _, _, st = fused.forward(dev(sc.pws), dev(sc.shs), dev(sc.alphas).reshape(-1, 1), dev(sc.scales), dev(sc.rots),
                             Camera.from_scene(cam), need_grad=True)
rg = host(st.ranges); ids = host(st.gaussian_ids()); masks = host(st.block_masks())
gx = (cam.width + 15) // 16
tile_of = np.repeat(np.arange(rg.shape[0]), rg[:, 1] - rg[:, 0])
for i, g in enumerate(rows):
    a = float(sc.alphas[g])
    listed = {int(tile_of[p]): int(masks[p]) for p in np.nonzero(ids == g)[0]}
    u, c = d_us[i], d_ci[i]
    r = int(np.ceil(3 * np.sqrt(max(d_c2[i][0], d_c2[i][2])))) + 1
    tx0, tx1 = int((u[0] - r) // 16), int((u[0] + r) // 16)
    ty0, ty1 = int((u[1] - r) // 16), int((u[1] + r) // 16)
    for ty in range(max(ty0, 0), min(ty1, 67) + 1):
        for tx in range(max(tx0, 0), min(tx1, 119) + 1):
            py, px = np.meshgrid(ty * 16 + np.arange(16.0), tx * 16 + np.arange(16.0), indexing="ij")
            dx = u[0] - px; dy = u[1] - py
            m = np.maximum(c[0] * dx * dx + c[2] * dy * dy + 2 * c[1] * dx * dy, 0)
            ap_ = np.minimum(a * np.exp(-0.5 * m), 0.99)
            hit = (ap_ >= 0.002) & (py < cam.height) & (px < cam.width)
            true_m = 0
            for k in range(4):
                if hit[8 * (k >> 1):8 * (k >> 1) + 8, 8 * (k & 1):8 * (k & 1) + 8].any():
                    true_m |= 1 << k
            dm = listed.get(ty * gx + tx)
            flag = "" if (dm is not None and (true_m & ~dm) == 0) or (dm is None and true_m == 0) else "   <-- MISSING"
            print("   gaussian %d tile (%d,%d): contributing blocks %s, device mask %s, max alpha' %.5f, pixels hit %d%s"
                  % (g, tx, ty, bin(true_m), "not listed" if dm is None else bin(dm), ap_.max(), int(hit.sum()), flag))
