"""CPU oracle pieces for the IO / formats row.

TEST INFRASTRUCTURE ONLY (imported by ``tests/``).  The file readers themselves are host code of
the product and are checked directly against fixture ``tests/golden/g9_io.npz`` (outputs of the
reference's ``gau_io.py`` / ``read_write_model.py`` on the byte strings stored in the fixture); what
needs an independent restatement is the one device kernel of this row:

* ``nn_sqdist``: squared distance to the nearest other point == the second column of
  ``faiss.IndexFlatL2(3).search(pws, 2)`` that read_write_model.py:216-220 clips into the initial scale.
  faiss (pinned only as ``faiss-gpu`` without version in the reference's requirements.txt) is absent
  here; IndexFlatL2 is an exact brute-force index, so the result is defined by the metric itself.
  Computed in float64, blockwise.  ``faiss_flat_l2_second`` restates the float32 formula faiss evaluates it with.
"""
import numpy as np


def nn_sqdist(points, block=2048):
    p = np.asarray(points, np.float64)
    n = p.shape[0]
    out = np.full(n, np.inf)
    sq = (p * p).sum(1)
    for a in range(0, n, block):
        pa = p[a:a + block]
        d = ((pa[:, None, :] - p[None, :, :]) ** 2).sum(-1) if n <= 4096 else \
            np.maximum(sq[a:a + block, None] + sq[None, :] - 2 * pa @ p.T, 0)
        d[np.arange(pa.shape[0]), np.arange(a, a + pa.shape[0])] = np.inf
        out[a:a + block] = d.min(1)
    return out


def faiss_flat_l2_second(points, block=1024):
    """The value reference read_write_model.py:219-222 actually reads: column 1 of
    ``faiss.IndexFlatL2(3).search(pws, 2)`` -- the SECOND smallest squared L2 distance of every row, the query itself
    included -- in faiss's own float32 arithmetic for more than 20 queries (faiss/utils/distances.cpp,
    ``exhaustive_L2sqr_blas``, ``distance_compute_blas_threshold`` = 20; unchanged between faiss 1.5 and 1.8):

        norms[i] = sum_k x[i,k]^2              (fvec_norms_L2sqr, float32)
        ip       = x y^T                        (sgemm, float32)
        dis[i,j] = norms_x[i] + norms_y[j] - 2 ip[i,j];   if (dis < 0) dis = 0

    faiss is absent from this image and pinned without a version by the reference (requirements.txt: ``faiss-gpu``),
    so this is a restatement of its PUBLISHED formula, not an output of the library: the summation order inside
    sgemm is BLAS's own.  tests/test_io_spec_fixtures.py bounds it against the exact metric, tests/test_gpu_io.py
    bounds the HIP kernel against both."""
    x = np.ascontiguousarray(points, np.float32)
    n = x.shape[0]
    norms = np.zeros(n, np.float32)
    for k in range(x.shape[1]):
        norms += x[:, k] * x[:, k]
    out = np.zeros(n, np.float32)
    for a in range(0, n, block):
        xa = x[a:a + block]
        ip = xa @ x.T                                           # float32 sgemm
        dis = norms[a:a + block, None] + norms[None, :] - np.float32(2) * ip
        dis = np.maximum(dis, np.float32(0))
        if n < 2:
            out[a:a + block] = np.float32(np.inf)
        else:
            out[a:a + block] = np.partition(dis, 1, axis=1)[:, 1]
    return out


# ---- the viewer's preprocess shader (reference viewer/shaders/gau_prep.glsl), float64 ------------------
# PARITY UNPINNED: a GLSL compute shader cannot be executed in this image (no OpenGL context), so this is a
# restatement of the shader text only; its building blocks (covariance, SH basis) are the ones pinned by
# G1/G2 through oracle/gs_oracle.py.
_SH_C = [0.28209479177387814, -0.4886025119029199, 0.4886025119029199, -0.4886025119029199,
         1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396,
         -0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]                       # gau_prep.glsl:13-28


def viewer_prep(gs_data, V, P, focal):
    """-> (prep [N,12], depth [N], culled [N] bool); culled rows carry u = -100 and zeros elsewhere."""
    g = np.asarray(gs_data, np.float64)
    V = np.asarray(V, np.float64).reshape(4, 4)
    P = np.asarray(P, np.float64).reshape(4, 4)
    n, K = g.shape[0], g.shape[1] - 11
    prep = np.zeros((n, 12))
    pw = np.concatenate([g[:, 0:3], np.ones((n, 1))], 1)
    pc = pw @ V.T                                                      # gau_prep.glsl:184
    u = pc @ P.T
    depth = pc[:, 2].copy()
    u = u / u[:, 3:4]
    culled = (np.abs(u[:, :2]) > 1.3).any(1) | (np.abs(u[:, 2]) > 1.0)
    w, x, y, z = g[:, 3], g[:, 4], g[:, 5], g[:, 6]                    # computeCov3D, gau_prep.glsl:66-91
    R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], 1),
                  np.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], 1),
                  np.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1)], 1)
    M = R * g[:, None, 7:10]
    Sigma = M @ M.transpose(0, 2, 1)
    zc = pc[:, 2]
    J = np.zeros((n, 3, 3))                                            # computeCov2D, gau_prep.glsl:93-112
    J[:, 0, 0] = focal[0] / zc; J[:, 0, 2] = -(focal[0] * pc[:, 0]) / zc ** 2
    J[:, 1, 1] = focal[1] / zc; J[:, 1, 2] = -(focal[1] * pc[:, 1]) / zc ** 2
    T = J @ V[:3, :3]
    cov = T @ Sigma @ T.transpose(0, 2, 1)
    c00, c01, c11 = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = c00 * c11 - c01 * c01
    culled |= det == 0
    cam = np.linalg.inv(V)[:3, 3]
    d = g[:, 0:3] - cam
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
    basis = [np.ones(n), y, z, x, xy, yz, 2 * zz - xx - yy, xz, xx - yy, y * (3 * xx - yy), xy * z,
             y * (4 * zz - xx - yy), z * (2 * zz - 3 * xx - 3 * yy), x * (4 * zz - xx - yy), z * (xx - yy),
             x * (xx - 3 * yy)]
    col = np.full((n, 3), 0.5)
    for c in range(K // 3):
        col += (_SH_C[c] * basis[c])[:, None] * g[:, 11 + 3 * c: 14 + 3 * c]
    with np.errstate(divide="ignore", invalid="ignore"):
        prep[:, 0:3] = u[:, :3]
        prep[:, 3] = c11 / det; prep[:, 4] = -c01 / det; prep[:, 5] = c00 / det
        prep[:, 6:9] = col
        prep[:, 9] = 3 * np.sqrt(c00); prep[:, 10] = 3 * np.sqrt(c11)
        prep[:, 11] = g[:, 10]
    prep[culled] = 0
    prep[culled, 0:3] = -100
    return prep, depth, culled
