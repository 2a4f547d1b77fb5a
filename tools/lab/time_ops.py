#!/usr/bin/env python3
"""HIP-event timing of single ops of the seven-op surface at the bench size (1 M Gaussians, SH degree 3)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easygaussiansplatting_amd import gsplatcu as gsc, scene as S

sc = S.big_scene()
dev = torch.device("cuda", 0)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
pws, shs, twc = t(sc.pws), t(sc.shs), t(sc.cam.twc)


def timed(fn, reps=200, warm=300):
    for _ in range(warm):
        fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for rnd in range(2):
    print("sh2Color calc_J=True  %.1f us   calc_J=False %.1f us" % (
        timed(lambda: gsc.sh2Color(shs, pws, twc, True)), timed(lambda: gsc.sh2Color(shs, pws, twc, False))), flush=True)
