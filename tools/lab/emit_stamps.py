"""Per-workgroup stamps of k_bin_emit on a scene.  Needs an INSTRUMENTED build (not the product): in a copy of the tree
(ab/inst, git-ignored) k_bin_emit stores wall_clock64 at its start / after the slot loop / at its end and the span into
`__device__ unsigned long long g_dbg[4 * 16384]`, read back by an extra `extern "C" int egs_dbg_read(void* host)`
(hipMemcpyFromSymbol).  Run from that copy:  python emit_stamps.py skewed|iid|skewed_reset   (docs/LAB.md, round 5)"""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
from easygaussiansplatting_amd import _lib, fused, scene as S
from easygaussiansplatting_amd.function import Camera
name = sys.argv[1]
lib = _lib.load()
sc = S.big_scene(1_000_000, 1920, 1080, 48) if name == "iid" else S.skewed_scene(reset_alpha=(name == "skewed_reset"))
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
cam = Camera.from_scene(sc.cam)
P = [dev(sc.pws), dev(sc.shs), dev(sc.alphas).reshape(-1, 1), dev(sc.scales), dev(sc.rots)]
with torch.no_grad():
    for _ in range(4):
        img, _, st = fused.forward(*P, cam)
    torch.cuda.synchronize()
buf = np.zeros(4 * 16384, np.uint64)
f = lib.egs_dbg_read; f.argtypes = [C.c_void_p]; f.restype = C.c_int
assert f(buf.ctypes.data) == 0
nb = (sc.n + 255) // 256
d = buf.reshape(-1, 4)[:nb].astype(np.int64)
t0, t1, t2, span = d[:, 0], d[:, 1], d[:, 2], d[:, 3]
base = t0.min()
print(name, "workgroups", nb, "kernel span %.1f us" % ((t2.max() - base) / 100.0), "slots", span.sum())
dur = (t2 - t0) / 100.0; slot = (t1 - t0) / 100.0; big = (t2 - t1) / 100.0
print(" duration us: median %.1f p90 %.1f p99 %.1f max %.1f ; slot-loop max %.1f ; big-rect part max %.1f median %.2f" % (
    np.median(dur), np.percentile(dur, 90), np.percentile(dur, 99), dur.max(), slot.max(), big.max(), np.median(big)))
print(" start times us: median %.1f p90 %.1f max %.1f" % (np.median((t0 - base) / 100.0), np.percentile((t0 - base) / 100.0, 90), ((t0 - base) / 100.0).max()))
o = np.argsort(-dur)[:8]
for i in o:
    print("  wg %5d start %.1f dur %.1f slot-loop %.1f big %.1f span %d" % (i, (t0[i] - base) / 100.0, dur[i], slot[i], big[i], span[i]))
o = np.argsort(-(t2 - base))[:5]
for i in o:
    print("  last-to-finish wg %5d start %.1f end %.1f span %d big %.1f" % (i, (t0[i] - base) / 100.0, (t2[i] - base) / 100.0, span[i], big[i]))
print(" corr(dur, span) %.2f ; sum of durations %.0f us ; span of top 1%% workgroups %.2f of all slots" % (
    np.corrcoef(dur, span)[0, 1], dur.sum(), np.sort(span)[-nb // 100:].sum() / max(1, span.sum())))
