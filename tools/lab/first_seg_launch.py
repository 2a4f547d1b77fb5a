"""Which launch pays the one-off 50 ms the first time a process takes the segment path (tools/lab/densify_first_step.py)?
Forward-only renders with the segment path forced, each waited for, then one forward + backward; per-kernel HIP-event
times of the first and the second of each."""
import ctypes
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from easygaussiansplatting_amd import _lib, fused, scene as S          # noqa: E402
from easygaussiansplatting_amd.function import Camera, GSFunction      # noqa: E402
from tools.benchlib import parse_report                                # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.load()
sc = S.big_scene(300_000, 1920, 1080, 48)
cam = Camera.from_scene(sc.cam, dev)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
P = [t(sc.pws), t(sc.shs), t(sc.alphas).reshape(-1, 1).clone(), t(sc.scales), t(sc.rots)]
us0 = torch.zeros((sc.n, 2), device=dev, requires_grad=True)
dl = torch.ones((3, 1080, 1920), device=dev) / (3 * 1920 * 1080)


def table():
    lib.egs_prof_enable(0)
    need = lib.egs_prof_report(None, 0)
    buf = ctypes.create_string_buffer(need + 16)
    lib.egs_prof_report(buf, need + 16)
    rep = parse_report(buf.value.decode())
    lib.egs_prof_reset()
    return {k: round(tot * 1e3) for k, (c, tot) in rep.items() if tot * 1e3 >= 300}


def timed(label, fn):
    torch.cuda.synchronize()
    lib.egs_prof_set_filter(None); lib.egs_prof_reset(); lib.egs_prof_enable(1)
    w0 = time.perf_counter()
    fn()
    w1 = time.perf_counter()
    torch.cuda.synchronize()
    w2 = time.perf_counter()
    print("%-44s host %.2f ms, done after %.2f ms; kernels >= 0.3 ms (us): %s" % (label, (w1 - w0) * 1e3, (w2 - w0) * 1e3, table()))


def fwd():
    with torch.no_grad():
        fused.forward(*P, cam, need_grad=True)


def step():
    for p in P:
        p.requires_grad_(True); p.grad = None
    us0.grad = None
    img, _ = GSFunction.apply(*P, us0, cam)
    img.backward(dl)


fused.SEGMENTS = "0"
for i in range(3):
    timed("unsplit forward %d" % i, fwd)
for i in range(2):
    timed("unsplit forward+backward %d" % i, step)
fused.SEGMENTS = "1"
for i in range(3):
    timed("segment path forward %d" % i, fwd)
for i in range(3):
    timed("segment path forward+backward %d" % i, step)
