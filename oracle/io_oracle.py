"""CPU oracle pieces for the IO / formats row.

TEST INFRASTRUCTURE ONLY (imported by ``tests/``).  The file readers themselves are host code of
the product and are checked directly against fixture ``tests/golden/g9_io.npz`` (outputs of the
reference's ``gau_io.py`` / ``read_write_model.py`` on the byte strings stored in the fixture); what
needs an independent restatement is the one device kernel of this row:

* ``nn_sqdist``: squared distance to the nearest other point == the second column of
  ``faiss.IndexFlatL2(3).search(pws, 2)`` that read_write_model.py:216-220 clips into the initial scale.
  faiss (pinned only as ``faiss-gpu`` without version in the reference's requirements.txt) is absent
  here; IndexFlatL2 is an exact brute-force index, so the result is defined by the metric itself.
  Computed in float64, blockwise.
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
