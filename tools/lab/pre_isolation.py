"""What does k_preprocess_fwd cost by itself, and what does it inherit from the kernel in front of it?  (round 6: in the
step it takes 92-95 us for 341 MB -- 0.46 of the HBM peak -- while tools/lab/ubench_rows.hip reads the same rows at
5.8 TB/s whatever the access pattern.)  Per-kernel averages (HIP events, egs_prof_*) of
  A  the full step (forward + backward: k_preprocess_fwd runs behind k_preprocess_bwd's 236 MB of dirty gradient rows)
  B  forward-only renders of the training instance (behind k_draw: 33 MB of dirty pixels)
  C  B with a 236-MB fill in front of every render (dirty lines of somebody else's, as in A)
  D  B with 20 us of idle stream in front (a device sleep is not available: an empty 1-thread kernel chain)"""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from easygaussiansplatting_amd import _lib, fused, scene as S          # noqa: E402
from easygaussiansplatting_amd.function import Camera, GSFunction      # noqa: E402
from tools.benchlib import parse_report                                # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.load()
sc = S.big_scene(1_000_000, 1920, 1080, 48)
cam = Camera.from_scene(sc.cam, dev)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
P = dict(pws=t(sc.pws), shs=t(sc.shs), alphas=t(sc.alphas).reshape(-1, 1).clone(), scales=t(sc.scales), rots=t(sc.rots))
us0 = torch.zeros((sc.n, 2), device=dev, requires_grad=True)
dl = torch.from_numpy(S.normal(1, 77, (3, 1080, 1920)).astype(np.float32)).to(dev) / (3 * 1920 * 1080)
junk = torch.empty(59 * sc.n, dtype=torch.float32, device=dev)


def table(fn, reps=12, warm=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    lib.egs_prof_set_filter(None); lib.egs_prof_reset(); lib.egs_prof_enable(1)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    lib.egs_prof_enable(0)
    need = lib.egs_prof_report(None, 0)
    buf = ctypes.create_string_buffer(need + 16)
    lib.egs_prof_report(buf, need + 16)
    rep = parse_report(buf.value.decode())
    lib.egs_prof_reset()
    return {k: round(tot / c * 1e3, 1) for k, (c, tot) in rep.items()}


def full_step():
    for p in P.values():
        p.requires_grad_(True); p.grad = None
    us0.grad = None
    with fused.deferred() as d:
        img, _ = GSFunction.apply(P["pws"], P["shs"], P["alphas"], P["scales"], P["rots"], us0, cam)
        img.backward(dl)
        d.commit()


def fwd_only():
    with torch.no_grad(), fused.deferred() as d:
        fused.forward(P["pws"].detach(), P["shs"].detach(), P["alphas"].detach(), P["scales"].detach(),
                      P["rots"].detach(), cam, need_grad=True)
        d.commit()


def fwd_after_fill():
    junk.fill_(1.0)
    fwd_only()


keys = ("k_preprocess_fwd", "k_radix_hist", "k_radix_scatter", "k_bin_scan_partials", "k_bin_scan_apply", "k_bin_emit",
        "k_tile_ranges", "k_draw", "k_draw_bwd", "k_preprocess_bwd")
for name, fn in (("A full step", full_step), ("B forward only", fwd_only), ("C forward after a 236-MB fill", fwd_after_fill)):
    tb = table(fn)
    print("%-32s" % name, "  ".join("%s %.1f" % (k.replace("k_", ""), tb[k]) for k in keys if k in tb))
