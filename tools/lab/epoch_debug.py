"""Per-step anatomy of bench.py's epoch_pattern leg around reset_alpha: step time, which draw path every render took
(from the per-kernel table of that step) and what the host's hint words said when the step was enqueued."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from easygaussiansplatting_amd import _lib, fused, scene as S          # noqa: E402
from easygaussiansplatting_amd.function import Camera, render          # noqa: E402
from easygaussiansplatting_amd.trainer import Trainer                  # noqa: E402
from tools.benchlib import parse_report                                # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.load()
sc = S.skewed_scene()
cams = [Camera.from_scene(c, dev) for c in S.ring_cameras(sc.cam, 8)]
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
with torch.no_grad():
    P = [t(sc.pws), t(sc.shs), t(sc.alphas), t(sc.scales), t(sc.rots)]
    gts = [render(*P, c)[0].clone() for c in cams]
    del P
start = S.skewed_scene()
start.pws[:] = start.pws + 0.004 * S.normal(11, 1, start.pws.shape).astype(np.float32)
tr = Trainer(start, cams, gts, max_steps=3000, scene_size=8.0, seed=1)
rng = np.random.default_rng(0)


def epoch(label, detail):
    for v in rng.permutation(8):
        n = int(tr.params["pws"].shape[0])
        hint = fused.seg_hint(dev, (n, sc.cam.width, sc.cam.height))
        if detail:
            lib.egs_prof_set_filter(None); lib.egs_prof_reset(); lib.egs_prof_enable(1)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        tr.step([int(v)], sync=False)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        if detail:
            lib.egs_prof_enable(0)
            need = lib.egs_prof_report(None, 0)
            buf = ctypes.create_string_buffer(need + 16)
            lib.egs_prof_report(buf, need + 16)
            rep = parse_report(buf.value.decode())
            draw = {k: round(tot * 1e3) for k, (c, tot) in rep.items() if k.startswith("k_draw") or k.startswith("k_seg")}
            print("%-28s view %d  %.3f ms  hint(longest, walk) at enqueue %s  after %s  %s"
                  % (label, v, ms, hint, fused.seg_hint(dev, (n, sc.cam.width, sc.cam.height)), draw))
        else:
            print("%-28s view %d  %.3f ms" % (label, v, ms))


epoch("first sight", False)
epoch("with history", False)
epoch("with history (detail)", True)
tr.reset_alpha()
epoch("after reset_alpha 1", True)
epoch("after reset_alpha 2", True)
epoch("after reset_alpha 3", True)
