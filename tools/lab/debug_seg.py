"""Dump and check the work items k_seg_plan wrote for a small scene (debug aid for the segment path)."""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from easygaussiansplatting_amd import _lib, fused, scene as S
from easygaussiansplatting_amd.function import Camera, GSFunction

L, MIN = int(sys.argv[1]), int(sys.argv[2])
lib = _lib.load()
_lib.check(lib.egs_seg_config(L, MIN, None))
fused.SEGMENTS = "1"
W, H = 320, 240
sc = S.small_scene(60_000, W, H, 12, seed=5)
sc.scales[:] = sc.scales * 2.2
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
cam = Camera.from_scene(sc.cam)
P = dict(pws=dev(sc.pws), shs=dev(sc.shs), alphas=dev(sc.alphas).reshape(-1, 1), scales=dev(sc.scales), rots=dev(sc.rots))
for p in P.values():
    p.requires_grad_(True)
dl = dev(S.normal(3, 21, (3, H, W)).astype(np.float32) / (3 * H * W))
for it in range(2):
    us0 = torch.zeros((sc.n, 2), device="cuda", requires_grad=True)
    img, mask = GSFunction.apply(P["pws"], P["shs"], P["alphas"], P["scales"], P["rots"], us0, cam)
    st = None
    import gc
    for o in gc.get_objects():
        if isinstance(o, fused.FusedState) and o.seg is not None:
            st = o
    img.backward(dl)
    torch.cuda.synchronize()
    T = (W // 16) * ((H + 15) // 16)
    ws = st.seg.cpu().numpy().view(np.int32)
    hdr = ws[:16]
    Tp = (T + 63) // 64 * 64
    o = 64 + 4096        # header + spare words + the plan's global bins (csrc/egs_raster.h SEG_PLAN_WORDS)
    seg_base = ws[o:o + T]; o += Tp
    walk = ws[o:o + T]; o += Tp
    items3 = ws[o:o + T]; o += Tp
    tmp = ws[o:o + T]; o += Tp
    tmp2 = ws[o:o + T]; o += Tp
    print("render", it, "hdr", hdr[:8])
    rg = st.ranges.cpu().numpy(); n = rg[:, 1] - rg[:, 0]
    nseg = (n + L - 1) // L
    split = n > MIN
    assert ((seg_base >= 0) == split).all(), "seg_base vs split"
    # slots disjoint
    b = seg_base[split]; e = b + nseg[split]
    order = np.argsort(b)
    assert (b[order][1:] >= e[order][:-1]).all(), "slots overlap"
    assert hdr[2] == nseg[split].sum(), ("slots", hdr[2], nseg[split].sum())
    # items1: what k_seg_plan wrote (n_plan items), then the segments COMPOSE walked itself and appended for the backward launch
    n1 = hdr[0]
    items1 = ws[o:o + n1].view(np.uint32)
    tile = items1 & 0x7FFFF; sg = (items1 >> 19) & 0x7FF; kind = items1 >> 30
    d = tile[kind == 0]
    assert np.array_equal(np.sort(d), np.nonzero(~split)[0]), "direct items"
    i3 = items3[:hdr[1]].view(np.uint32)
    assert np.array_equal(np.sort(i3 & 0x7FFFF), np.nonzero(split)[0]), "items3"
    nspec = np.zeros(T, np.int64); nspec[i3 & 0x7FFFF] = (i3 >> 19) & 0x7FF
    cont = st.contrib.cpu().numpy()
    gx = W // 16
    wt = np.array([cont[(t // gx) * 16:(t // gx) * 16 + 16, (t % gx) * 16:(t % gx) * 16 + 16].max() for t in range(T)])
    assert np.array_equal(wt, walk), ("walk", np.nonzero(wt != walk)[0][:10], wt[:5], walk[:5])
    need = (walk + L - 1) // L
    bad = 0
    for t in np.nonzero(split)[0]:
        s_ = np.sort(sg[(tile == t) & (kind == 1)])
        # the planned SPEC items 0 .. nspec-1, then whatever COMPOSE walked beyond them: every walked segment is there
        want = np.arange(max(nspec[t], need[t]))
        if not (np.array_equal(s_[:nspec[t]], np.arange(nspec[t])) and set(np.arange(need[t])) <= set(s_) and len(s_) == len(set(s_))):
            bad += 1
            if bad < 5: print("   tile", t, "walk", walk[t], "need", need[t], "nspec", nspec[t], "got", s_)
    print("  items ok: direct", (kind == 0).sum(), "spec", (kind == 1).sum(), "compose", hdr[1], "max nspec", nspec.max(),
          "appended by compose", int((kind == 1).sum() - nspec[split].sum()), "bad tiles", bad)
