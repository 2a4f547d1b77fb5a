"""The oracle's per-tile backward pass (``O.draw_backward``) over MANY tiles on the host's cores.

Test infrastructure: at 1 M Gaussians / 1920x1080 the serial oracle needs ~12 minutes for all 8160 tiles; the GPU box has
hundreds of host cores, so the all-tile gradient comparison of tests/test_gpu_parity.py hands the tiles to a pool of
worker processes (``spawn``: the parent holds a HIP context, which must not be forked).  Inputs travel as ``.npy``
files in a temporary directory (memory-mapped by the workers), results come back sparse -- the rows of the Gaussians
a worker's tiles list -- and are summed here in float64.  The arithmetic is ``O.draw_backward``'s, untouched."""
import multiprocessing as mp
import os
import shutil
import sys
import tempfile

import numpy as np

_NAMES = ("ranges", "gsid", "us", "cinv2ds", "alphas", "colors", "contrib", "final_tau", "dl")


def _worker(job):
    tmp, width, height, tiles, near_margin, near_u_ulps, behind, repo = job
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from oracle import gs_oracle as O
    a = {k: np.load(os.path.join(tmp, k + ".npy"), mmap_mode="r") for k in _NAMES}
    n = a["us"].shape[0]
    near = np.zeros(n, bool)
    cabs = np.zeros((n, 3))
    cbeh = np.zeros((n, 3)) if behind else None
    g = O.draw_backward(width, height, a["ranges"], a["gsid"], a["us"], a["cinv2ds"], a["alphas"], a["colors"],
                        a["contrib"], a["final_tau"], a["dl"], None, O.POLICY_G, tiles=tiles, near_out=near,
                        near_margin=near_margin, near_u_ulps=near_u_ulps, abs_out=cabs, behind_out=cbeh)
    rg = a["ranges"]
    ids = np.unique(np.concatenate([np.asarray(a["gsid"][rg[t, 0]:rg[t, 1]]) for t in tiles] or [np.zeros(0, np.int64)]))
    ids = ids.astype(np.int64)
    return ids, g[0][ids], g[1][ids], g[2][ids], g[3][ids], near[ids], cabs[ids], (cbeh[ids] if behind else None)


def draw_backward_tiles(width, height, ranges, gsid, us, cinv2ds, alphas, colors, contrib, final_tau, dl, tiles=None,
                        near_margin=1e-4, procs=None, chunks_per_proc=6, near_u_ulps=0.0, behind=False):
    """-> (dus[N,2], dcinv[N,3], dalpha[N], dcolor[N,3], near[N], dcolor_abs[N,3]) over ``tiles`` (default: all),
    POLICY_G; dcolor_abs = the sum of the absolute pixel terms of dcolor (``abs_out`` of O.draw_backward); with
    ``behind`` a seventh array: ``behind_out`` of O.draw_backward (twice the time)."""
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ranges = np.asarray(ranges); gsid = np.asarray(gsid)
    n = np.asarray(us).shape[0]
    tiles = np.arange(ranges.shape[0]) if tiles is None else np.asarray(tiles)
    lens = (ranges[tiles, 1] - ranges[tiles, 0]).astype(np.int64)
    tiles = tiles[lens > 0][np.argsort(-lens[lens > 0], kind="stable")]         # longest lists first
    procs = max(1, min(procs or (os.cpu_count() or 1), 96, len(tiles)))
    nchunk = max(1, min(len(tiles), procs * chunks_per_proc))
    chunks = [tiles[i::nchunk] for i in range(nchunk)]                          # dealt round-robin: balanced
    arrays = dict(ranges=ranges, gsid=gsid, us=np.asarray(us, np.float64), cinv2ds=np.asarray(cinv2ds, np.float64),
                  alphas=np.asarray(alphas, np.float64).reshape(-1), colors=np.asarray(colors, np.float64),
                  contrib=np.asarray(contrib), final_tau=np.asarray(final_tau, np.float64),
                  dl=np.asarray(dl, np.float64))
    need = sum(v.nbytes for v in arrays.values())
    # /dev/shm when it has room (a container's default is 64 MB), else the ordinary temporary directory
    shm = "/dev/shm" if (os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 2 * need + (64 << 20)) else None
    tmp = tempfile.mkdtemp(prefix="egs_oracle_", dir=shm)
    try:
        for k, v in arrays.items():
            np.save(os.path.join(tmp, k + ".npy"), v)
        jobs = [(tmp, width, height, c, near_margin, near_u_ulps, behind, repo) for c in chunks]
        dus = np.zeros((n, 2)); dcinv = np.zeros((n, 3)); dalpha = np.zeros(n); dcolor = np.zeros((n, 3))
        near = np.zeros(n, bool)
        cabs = np.zeros((n, 3))
        cbeh = np.zeros((n, 3))
        if procs == 1:
            results = map(_worker, jobs)
            pool = None
        else:
            pool = mp.get_context("spawn").Pool(procs)
            results = pool.imap_unordered(_worker, jobs)
        try:
            for ids, a, b, c, d, nr, ca, cb in results:
                np.add.at(dus, ids, a); np.add.at(dcinv, ids, b); np.add.at(dalpha, ids, c); np.add.at(dcolor, ids, d)
                np.add.at(cabs, ids, ca)
                if cb is not None:
                    np.add.at(cbeh, ids, cb)
                near[ids] |= nr
        finally:
            if pool is not None:
                pool.close(); pool.join()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return (dus, dcinv, dalpha, dcolor, near, cabs, cbeh) if behind else (dus, dcinv, dalpha, dcolor, near, cabs)
