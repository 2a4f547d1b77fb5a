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
    o = 64
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
    # items1
    n1 = hdr[0]
    slots_total = (st.seg.numel() - 4 * (64 + 5 * Tp + 2 * (T + 0 + 64))) // (256 * 6 * 4 + 8)
    items1 = ws[o:o + n1].view(np.uint32)
    tile = items1 & 0x7FFFF; sg = (items1 >> 19) & 0x7FF; kind = items1 >> 30
    d = tile[kind == 0]
    assert np.array_equal(np.sort(d), np.nonzero(~split)[0]), "direct items"
    i3 = items3[:hdr[1]].view(np.uint32)
    assert np.array_equal(np.sort(i3 & 0x7FFFF), np.nonzero(split)[0]), "items3"
    nspec = np.zeros(T, np.int64); nspec[i3 & 0x7FFFF] = (i3 >> 19) & 0x7FF
    for t in np.nonzero(split)[0]:
        s_ = np.sort(sg[(tile == t) & (kind == 1)])
        assert np.array_equal(s_, np.arange(nspec[t])), (t, s_, nspec[t])
    print("  forward items ok: direct", (kind == 0).sum(), "spec", (kind == 1).sum(), "compose", hdr[1], "max nspec", nspec.max())
    cont = st.contrib.cpu().numpy()
    gx = W // 16
    wt = np.array([cont[(t // gx) * 16:(t // gx) * 16 + 16, (t % gx) * 16:(t % gx) * 16 + 16].max() for t in range(T)])
    assert np.array_equal(wt, walk), ("walk", np.nonzero(wt != walk)[0][:10], wt[:5], walk[:5])
    nb = hdr[5]
    o2 = o + T + slots_total + 64
    # recompute slots exactly as seg_carve does
    bytes_ = st.seg.numel()
    def wsb(slots): return 4 * (16 + 48 + 5 * Tp + 2 * (T + slots + 64)) + slots * 256 * 6 * 4 + 1024
    slots = (bytes_ - wsb(0)) // (256 * 6 * 4 + 8)
    o2 = o + T + slots + 64
    itemsB = ws[o2:o2 + nb].view(np.uint32)
    tb = itemsB & 0x7FFFF; sb_ = (itemsB >> 19) & 0x7FF; kb = itemsB >> 30
    need = (walk + L - 1) // L
    print("   tmp2[:12]", tmp2[:12], "need[:12]", need[:12], "nseg[:12]", nseg[:12], "walk[:12]", walk[:12])
    print("   tmp[:6] bins", (tmp[:6].view(np.uint32) >> 20), "rank", tmp[:6].view(np.uint32) & 0xFFFFF, "itemsB[:8]", [hex(x) for x in itemsB[:8]])
    bad = 0
    for t in range(T):
        if split[t]:
            s_ = np.sort(sb_[(tb == t) & (kb == 1)])
            if not np.array_equal(s_, np.arange(need[t])):
                bad += 1
                if bad < 5: print("   tile", t, "walk", walk[t], "need", need[t], "got", s_)
        else:
            assert ((tb == t) & (kb == 0)).sum() == 1, t
    print("  walk ok, itemsB", nb, "expected", need[split].sum() + (~split).sum(), "bad tiles", bad)
