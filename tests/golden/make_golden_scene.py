"""The seeded 256-Gaussian stage scene shared by tests/golden/make_golden.py
(fixtures G1/G2) and the GPU parity tests.  No reference import here."""
import numpy as np

from easygaussiansplatting_amd import scene as S


def stage_scene(n=256, seed=7):
    """n random Gaussians (SH degree 3) around the frustum of a 640x480 camera:
    ~15% lie outside the field of view / behind the camera."""
    u = S.uniform01(seed, 11, (n, 3))
    pws = np.stack([-6 + 12 * u[:, 0], -4 + 8 * u[:, 1], -1.5 + 9.5 * u[:, 2]], 1)
    q = S.normal(seed, 12, (n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    s = np.exp(np.log(0.01) + S.uniform01(seed, 13, (n, 3)) * (np.log(0.5) - np.log(0.01)))
    a = 0.05 + 0.94 * S.uniform01(seed, 14, (n,))
    sh = 0.3 * S.normal(seed, 15, (n, 48))
    th = 0.3
    Rcw = np.array([[np.cos(th), 0, -np.sin(th)], [0, 1, 0], [np.sin(th), 0, np.cos(th)]]) @ \
        np.array([[1, 0, 0], [0, np.cos(0.1), -np.sin(0.1)], [0, np.sin(0.1), np.cos(0.1)]])
    tcw = np.array([0.2, -0.1, 1.0])
    cam = S.Camera(640, 480, 500.0, 480.0, 320.0, 240.0, Rcw, tcw)
    f = np.float32
    return S.Scene(pws.astype(f), q.astype(f), s.astype(f), a.astype(f), sh.astype(f), cam)
