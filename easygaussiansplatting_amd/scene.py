"""Deterministic synthetic scenes for parity tests and bench.py.

The reference has no scene generator beyond ``get_example_gs()``
(reference gsplat/gau_io.py:159-183) and uses *unseeded* ``np.random.rand``
for the SH rest coefficients (reference backward_cpu.py:507).  Everything here
is derived from an in-repo counter-based RNG (splitmix64 over the element
index) so that the same bytes come out on any numpy version / any machine:
the golden fixtures under ``tests/golden`` were generated from these scenes.

Record layout follows the reference's ``gsdata_type`` (gau_io.py:7-12):
``pw f4[3], rot f4[4] (w,x,y,z), scale f4[3], alpha f4, sh f4[K]`` with
``sh[i, 3*c + rgb]``.
"""
from __future__ import annotations

import dataclasses
import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """One splitmix64 output step, vectorised on uint64 (wraps mod 2**64)."""
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def uniform01(seed: int, stream: int, shape) -> np.ndarray:
    """float64 uniforms in [0,1): element e of stream s is a pure function of
    (seed, s, e) -- no generator state, so sub-sampling a scene is stable."""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64)
        key = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream))
        bits = _splitmix64(ctr ^ key)
        bits = _splitmix64(bits + key)
    # 53 random mantissa bits
    return ((bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))).reshape(shape)


def normal(seed: int, stream: int, shape) -> np.ndarray:
    """Standard normals by Box-Muller on two independent uniform streams."""
    u1 = uniform01(seed, 2 * stream + 1000, shape)
    u2 = uniform01(seed, 2 * stream + 1001, shape)
    u1 = np.maximum(u1, 1e-300)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def gsdata_type(sh_dim: int):
    """Structured dtype of a Gaussian record == reference gau_io.py:7-12."""
    return [("pw", "<f4", (3,)), ("rot", "<f4", (4,)), ("scale", "<f4", (3,)),
            ("alpha", "<f4"), ("sh", "<f4", (sh_dim,))]


@dataclasses.dataclass
class Camera:
    """Pin-hole camera; field names follow reference gausplat_dataset.py:14-26."""
    width: int
    height: int
    fx: float
    fy: float
    cx: float
    cy: float
    Rcw: np.ndarray  # [3,3] float64
    tcw: np.ndarray  # [3]   float64

    @property
    def twc(self) -> np.ndarray:
        return np.linalg.inv(self.Rcw) @ (-self.tcw)


@dataclasses.dataclass
class Scene:
    pws: np.ndarray      # [N,3] f32
    rots: np.ndarray     # [N,4] f32 (w,x,y,z), unit norm
    scales: np.ndarray   # [N,3] f32
    alphas: np.ndarray   # [N]   f32
    shs: np.ndarray      # [N,K] f32
    cam: Camera

    @property
    def n(self) -> int:
        return self.pws.shape[0]

    def as_records(self) -> np.ndarray:
        return np.rec.fromarrays([self.pws, self.rots, self.scales, self.alphas, self.shs],
                                 dtype=gsdata_type(self.shs.shape[1]))

    def subsample(self, idx) -> "Scene":
        return Scene(self.pws[idx], self.rots[idx], self.scales[idx], self.alphas[idx],
                     self.shs[idx], self.cam)


def _make(seed, n, sh_dim, box, scale_lo, scale_hi, alpha_lo, alpha_hi, sh_sigma, cam) -> Scene:
    u = uniform01(seed, 1, (n, 3))
    lo = np.array([b[0] for b in box]); hi = np.array([b[1] for b in box])
    pws = lo + u * (hi - lo)
    q = normal(seed, 2, (n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    s = np.exp(np.log(scale_lo) + uniform01(seed, 3, (n, 3)) * (np.log(scale_hi) - np.log(scale_lo)))
    a = alpha_lo + uniform01(seed, 4, (n,)) * (alpha_hi - alpha_lo)
    sh = sh_sigma * normal(seed, 5, (n, sh_dim))
    f = np.float32
    return Scene(pws.astype(f), q.astype(f), s.astype(f), a.astype(f), sh.astype(f), cam)


def big_scene(n: int = 1_000_000, width: int = 1920, height: int = 1080, sh_dim: int = 48,
              seed: int = 0) -> Scene:
    """BASELINE configs[1]/[2]: 1 M Gaussians, 1920x1080, SH degree 3.

    All Gaussians are inside the frustum (depth 4..8) so the fov clamp of
    compute_cov_2d is inactive under every policy; alpha <= 0.99 so the 0.99
    clamp never binds."""
    cam = Camera(width, height, 1200.0, 1200.0, width / 2.0, height / 2.0,
                 np.eye(3), np.array([0.0, 0.0, 6.0]))
    return _make(seed, n, sh_dim, [(-4, 4), (-2.25, 2.25), (-2, 2)], 0.003, 0.03, 0.05, 0.99, 0.3, cam)


def skewed_scene(n: int = 1_500_000, width: int = 1920, height: int = 1080, sh_dim: int = 48, seed: int = 7,
                 reset_alpha: bool = False, fillers: int = 160) -> Scene:
    """A heavy-tailed scene of the shape a trained model has (BASELINE configs[4] is real data, absent here): what
    ``big_scene`` -- iid positions, every Gaussian <= ~4x4 tiles, lists 114..830 -- does not exercise.

    * positions: 62 % of the Gaussians in eight depth-clustered blobs (sigma 0.12 .. 1.2 world units, depths 4.3 .. 8:
      the densest puts > 10 k entries on its centre tiles and thousands of equal mm depth keys), the rest iid in
      ``big_scene``'s box;
    * scales: log-normal about 0.008 (sigma 0.85) with per-axis anisotropy 1/3 .. 3, clipped to [0.0015, 0.6]; on top,
      ``fillers`` Gaussians with scales 0.35 .. 1.6 -- 3-sigma rects of 30 x 30 tiles up to the whole screen (the
      row-walk paths of the binning, thousands of tiles per Gaussian);
    * opacity bimodal: half U(0.55, 0.99), half U(0.004, 0.12) (some below alpha_skip = 0.002 / 0.99 never blend);
      ``reset_alpha``: every opacity min(alpha, 0.01), the state right after the reference's ``reset_alpha``
      (gsmodel.py:320-324, train.py:77): nothing saturates, every tile walks its whole list.
    Same camera as ``big_scene``; every Gaussian in front of it (depth > 3.4)."""
    cam = Camera(width, height, 1200.0, 1200.0, width / 2.0, height / 2.0,
                 np.eye(3), np.array([0.0, 0.0, 6.0]))
    u = uniform01(seed, 1, (n, 3))
    box = np.array([[-4, 4], [-2.25, 2.25], [-2, 2]], np.float64)
    pws = box[:, 0] + u * (box[:, 1] - box[:, 0])
    # blobs: centre (x, y, z), sigma (x == y, z), share of the Gaussians
    blobs = [((-2.6, -1.2, -1.6), 0.12, 0.05, 0.07), ((1.9, 0.9, -1.0), 0.16, 0.08, 0.09),
             ((0.2, -0.3, 0.0), 0.30, 0.10, 0.10), ((-1.2, 1.3, 0.8), 0.22, 0.05, 0.07),
             ((2.9, -1.4, 1.4), 0.45, 0.20, 0.08), ((-3.0, 0.6, 1.9), 0.60, 0.15, 0.07),
             ((0.9, 1.5, -1.7), 0.15, 0.04, 0.06), ((0.0, 0.0, 1.0), 1.20, 0.30, 0.08)]
    pick = uniform01(seed, 6, (n,))
    g3 = normal(seed, 7, (n, 3))
    lo = 0.0
    for (cx_, cy_, cz_), sxy, sz, share in blobs:
        m = (pick >= lo) & (pick < lo + share)
        pws[m] = np.array([cx_, cy_, cz_]) + g3[m] * np.array([sxy, sxy, sz])
        lo += share
    pws[:, 2] = np.maximum(pws[:, 2], -2.6)            # depth = z + 6 > 3.4
    q = normal(seed, 2, (n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    base = np.exp(np.log(0.008) + 0.85 * normal(seed, 8, (n, 1)))
    aniso = np.exp((uniform01(seed, 3, (n, 3)) * 2.0 - 1.0) * np.log(3.0))
    s = np.clip(base * aniso, 0.0015, 0.6)
    if fillers > 0:                                    # every (n // fillers)-th Gaussian fills (a part of) the screen
        idx = (np.arange(fillers) * (n // fillers) + n // (2 * fillers)) % n
        fs = np.exp(np.log(0.35) + uniform01(seed, 9, (fillers, 1)) * (np.log(1.6) - np.log(0.35)))
        s[idx] = fs * aniso[idx] ** 0.5
        pws[idx, 2] = 0.5 + 1.5 * uniform01(seed, 10, (fillers,))
    ua = uniform01(seed, 4, (n,))
    strong = uniform01(seed, 11, (n,)) < 0.5
    a = np.where(strong, 0.55 + ua * 0.44, 0.004 + ua * 0.116)
    if reset_alpha:
        a = np.minimum(a, 0.01)
    sh = 0.3 * normal(seed, 5, (n, sh_dim))
    f = np.float32
    return Scene(pws.astype(f), q.astype(f), s.astype(f), a.astype(f), sh.astype(f), cam)


def small_scene(n: int = 10_000, width: int = 256, height: int = 256, sh_dim: int = 3,
                seed: int = 0) -> Scene:
    """BASELINE configs[0]: 10 k Gaussians, 256x256, SH degree 0."""
    cam = Camera(width, height, 256.0, 256.0, width / 2.0, height / 2.0,
                 np.eye(3), np.array([0.0, 0.0, 5.0]))
    return _make(seed, n, sh_dim, [(-2, 2), (-2, 2), (-2, 2)], 0.005, 0.05, 0.1, 0.99, 0.3, cam)


def ring_cameras(base: Camera, n_views: int = 8, radius: float = 6.0):
    """8-view config: cameras on a ring of `radius` around the
    origin looking inward, yaw k*360/n_views about the world y axis."""
    cams = []
    for k in range(n_views):
        th = 2.0 * np.pi * k / n_views
        c, s = np.cos(th), np.sin(th)
        # camera-from-world rotation for a yaw of th about +y; the camera
        # centre sits at twc = Rwc @ (0,0,-radius), optical axis through 0.
        Rcw = np.array([[c, 0.0, -s], [0.0, 1.0, 0.0], [s, 0.0, c]])
        tcw = np.array([0.0, 0.0, radius])
        cams.append(Camera(base.width, base.height, base.fx, base.fy, base.cx, base.cy, Rcw, tcw))
    return cams


def example_gs() -> Scene:
    """The reference's only fixture: 4 Gaussians (gau_io.py:159-183) with the
    camera of backward_cpu.py:516-526 (32x16, fx=fy=16)."""
    g = np.array([[0., 0., 0., 1., 0., 0., 0., 0.05, 0.05, 0.05, 1., 1.772484, -1.772484, 1.772484],
                  [1., 0., 0., 1., 0., 0., 0., 0.2, 0.05, 0.05, 1., 1.772484, -1.772484, -1.772484],
                  [0., 1., 0., 1., 0., 0., 0., 0.05, 0.2, 0.05, 1., -1.772484, 1.772484, -1.772484],
                  [0., 0., 1., 1., 0., 0., 0., 0.05, 0.05, 0.2, 1., -1.772484, -1.772484, 1.772484]],
                 dtype=np.float32)
    Rcw = np.array([[0.89699204, 0.06525223, 0.43720409],
                    [-0.04508268, 0.99739184, -0.05636552],
                    [-0.43974177, 0.03084909, 0.89759429]]).T
    tcw = np.array([1.03796196, 0.42017467, 4.67804612])
    cam = Camera(32, 16, 16.0, 16.0, 16.0, 8.0, Rcw, tcw)
    return Scene(g[:, 0:3].copy(), g[:, 3:7].copy(), g[:, 7:10].copy(), g[:, 10].copy(),
                 g[:, 11:14].copy(), cam)
