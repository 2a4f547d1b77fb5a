"""bench.py's epoch_pattern leg: the FIRST step after a densification took 51 ms (the seven after it 1.8 ms each): one
step in a hundred of the reference's loop (train.py:71-73), +0.5 ms per step amortized on a 1.7 ms step.  Where does it
go?  Host wall time and GPU time of the densification itself and of every step of the epoch after it, the allocator's
counters around each (a hipMalloc of a new size is a device-wide wait), and a cProfile of the first step."""
import cProfile
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from easygaussiansplatting_amd import _lib, dist_views, fused, function, scene as S   # noqa: E402
from easygaussiansplatting_amd.function import Camera, render          # noqa: E402
from easygaussiansplatting_amd.trainer import Trainer                  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.load()
SLOW = []


def wrap(mod, name):
    """record calls of mod.name that take the host more than 1 ms (the autograd thread is invisible to cProfile)"""
    f = getattr(mod, name)

    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            dt = (time.perf_counter() - t0) * 1e3
            if dt > 1.0:
                SLOW.append("%s.%s %.1f ms" % (getattr(mod, "__name__", mod), name, dt))
    setattr(mod, name, g)


for nm in ("backward", "forward", "commit", "sh_sink_for", "accumulation_targets"):
    wrap(fused, nm)
import types                                                             # noqa: E402
for nm, f in list(vars(dist_views.FactoredShGrad).items()):
    if isinstance(f, types.FunctionType) and not nm.startswith("__"):
        wrap(dist_views.FactoredShGrad, nm)
for nm, f in list(vars(function).items()):
    if isinstance(f, types.FunctionType):
        wrap(function, nm)
import ctypes                                                            # noqa: E402
from tools.benchlib import parse_report                                # noqa: E402
sc = S.skewed_scene()
cams = [Camera.from_scene(c, dev) for c in S.ring_cameras(sc.cam, 8)]
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
with torch.no_grad():
    P = [t(sc.pws), t(sc.shs), t(sc.alphas), t(sc.scales), t(sc.rots)]
    gts = [render(*P, c)[0].clone() for c in cams]
    del P
start = S.skewed_scene()
start.pws[:] = start.pws + 0.004 * S.normal(11, 1, start.pws.shape).astype(np.float32)
start.shs[:, :3] += 0.3 * S.normal(11, 2, (start.n, 3)).astype(np.float32)
tr = Trainer(start, cams, gts, max_steps=3000, scene_size=8.0, seed=1)
rng = np.random.default_rng(0)


def alloc():
    s = torch.cuda.memory_stats(dev)
    return (s["num_device_alloc"], s["num_device_free"], s["num_alloc_retries"], s["reserved_bytes.all.current"] >> 20,
            s["allocated_bytes.all.current"] >> 20)


def epoch(label, profile_first=False):
    for j, v in enumerate(rng.permutation(8)):
        a0 = alloc()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        w0 = time.perf_counter()
        e0.record()
        if profile_first and j == 0:
            lib.egs_prof_set_filter(None); lib.egs_prof_reset(); lib.egs_prof_enable(1)
            del SLOW[:]
            pr = cProfile.Profile()
            pr.enable()
        tr.step([int(v)], sync=False)
        if profile_first and j == 0:
            pr.disable()
        e1.record()
        w1 = time.perf_counter()
        torch.cuda.synchronize()
        w2 = time.perf_counter()
        a1 = alloc()
        print("%-22s view %d  host enqueue %.2f ms  until done %.2f ms  GPU events %.2f ms  device allocs +%d frees +%d retries +%d  "
              "reserved %d -> %d MiB (in use %d)" % (label, v, (w1 - w0) * 1e3, (w2 - w0) * 1e3, e0.elapsed_time(e1),
                                                     a1[0] - a0[0], a1[1] - a0[1], a1[2] - a0[2], a0[3], a1[3], a1[4]))
        if profile_first and j == 0:
            lib.egs_prof_enable(0)
            need = lib.egs_prof_report(None, 0)
            buf = ctypes.create_string_buffer(need + 16)
            lib.egs_prof_report(buf, need + 16)
            rep = parse_report(buf.value.decode())
            lib.egs_prof_reset()
            print("   kernels (us):", {k: round(tot * 1e3) for k, (c, tot) in rep.items()})
            print("   slow host calls:", SLOW)
            pstats.Stats(pr).sort_stats("cumulative").print_stats(12)


for rounds in range(2):
    epoch("first sight" if rounds == 0 else "with history")
    epoch("with history")
    a0 = alloc()
    torch.cuda.synchronize()
    w0 = time.perf_counter()
    rep = tr.densify()
    torch.cuda.synchronize()
    a1 = alloc()
    print("densify %s: %.2f ms  device allocs +%d frees +%d  reserved %d -> %d MiB" % (rep, (time.perf_counter() - w0) * 1e3,
                                                                                   a1[0] - a0[0], a1[1] - a0[1], a0[3], a1[3]))
    epoch("after densify %d" % rounds, profile_first=(rounds == 0))
    epoch("after densify %d, 2nd" % rounds)
