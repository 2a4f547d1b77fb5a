#!/usr/bin/env python3
"""Counterpart of the reference's ``forward_gpu.py``: the six-call inference sequence on the drop-in
``gsplatcu`` module, same camera, image written to a file instead of a matplotlib window.

    python examples/forward_gpu.py [--gs scene.npy|scene.ply] [--out image.png] [--policy gsplatcu|forward_cpu]

(The reference passes ``height, width`` swapped to ``computeCov2D`` at forward_gpu.py:53; the op's
declared order ``width, height`` (ext.cpp:44-52) is used here.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gs", help="the gs path (.npy record array or 3DGS .ply); default: the 4-Gaussian example")
    ap.add_argument("--out", default="forward_gpu.png")
    ap.add_argument("--policy", default="gsplatcu", choices=["gsplatcu", "forward_cpu"])
    a = ap.parse_args()

    import torch
    import gsplatcu as gsc                                   # the drop-in module of this repository
    from easygaussiansplatting_amd.gau_io import get_example_gs, load_gs

    if a.gs:
        print("Try to load %s ..." % a.gs)
        gs = load_gs(a.gs)
    else:
        print("not gs file.")
        gs = get_example_gs()
    gsc.set_policy(a.policy)

    # camera of forward_gpu.py:21-33 / forward_cpu.py:20-31
    tcw = np.array([1.03796196, 0.42017467, 4.67804612])
    Rcw = np.array([[0.89699204, 0.06525223, 0.43720409],
                    [-0.04508268, 0.99739184, -0.05636552],
                    [-0.43974177, 0.03084909, 0.89759429]]).T
    width, height = 979, 546
    fx, fy = 581.6273640151177, 578.140202494143
    cx, cy = width / 2, height / 2

    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).type(torch.float32).to("cuda")
    pws, rots, scales, alphas, shs = (dev(gs[k]) for k in ("pw", "rot", "scale", "alpha", "sh"))
    Rcw, tcw = dev(Rcw), dev(tcw)
    twc = torch.linalg.inv(Rcw) @ (-tcw)

    us, pcs, depths = gsc.project(pws, Rcw, tcw, fx, fy, cx, cy, False)                      # step 1
    cov3ds = gsc.computeCov3D(rots, scales, depths, False)[0]                                # step 2
    cov2ds = gsc.computeCov2D(cov3ds, pcs, Rcw, depths, fx, fy, width, height, False)[0]     # step 3
    colors = gsc.sh2Color(shs.reshape(shs.shape[0], -1), pws, twc, False)[0]                 # step 4
    cinv2ds, areas = gsc.inverseCov2D(cov2ds, depths, False)                                 # step 5
    image = gsc.splat(height, width, us, cinv2ds, alphas, depths, colors, areas)[0]
    image = image.to("cpu").numpy()

    from PIL import Image
    Image.fromarray((np.clip(image.transpose(1, 2, 0), 0, 1) * 255 + 0.5).astype(np.uint8)).save(a.out)
    print("wrote %s (%dx%d, %d Gaussians)" % (a.out, width, height, pws.shape[0]))


if __name__ == "__main__":
    main()
