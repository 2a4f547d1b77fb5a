#!/usr/bin/env python3
"""What k_draw_bwd walks on the bench scene (fused path, culled lists with exact block masks), measured from the
render's own state: per (tile, entry)
  reach  -- the entry passes the kernel's `todo` test: some block of its mask has idx < bmax[block];
  hit    -- some pixel of a reachable block blends it (idx < contrib[pixel] and power >= thr): kernel.cu:899,913;
and per (entry, block): tested / hit, plus the fraction of the 64 lanes that pass inside a hit block.
The gap reach - hit is what a per-entry "blended something" bit left behind by k_draw would save."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easygaussiansplatting_amd import fused, scene as S
from easygaussiansplatting_amd.function import Camera

W, H = 1920, 1080
sc = S.big_scene(1_000_000, W, H, 48)
dev = torch.device("cuda", 0)
cam = Camera.from_scene(sc.cam, dev)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
with torch.no_grad():
    img, mask, st = fused.forward(t(sc.pws), t(sc.shs), t(sc.alphas), t(sc.scales), t(sc.rots), cam, need_grad=True)
torch.cuda.synchronize()
P = st.patch_count()
ranges = st.ranges.long()
gx, gy = (W + 15) // 16, (H + 15) // 16
T = ranges.shape[0]
lens = ranges[:, 1] - ranges[:, 0]
tile_of = torch.repeat_interleave(torch.arange(T, device=dev), lens)          # [P]
idx = torch.arange(P, device=dev) - ranges[tile_of, 0]                         # forward index in the tile list
g = st.gaussian_ids().long()
masks = st.block_masks().long()
rec = st.rec
cont = torch.zeros((gy * 16, gx * 16), dtype=torch.long, device=dev)
cont[:H, :W] = st.contrib.long()
cont_t = cont.reshape(gy, 16, gx, 16).permute(0, 2, 1, 3).reshape(T, 16, 16)   # [T,16,16]
bmax = torch.stack([cont_t[:, 8 * (k >> 1):8 * (k >> 1) + 8, 8 * (k & 1):8 * (k & 1) + 8].amax((1, 2))
                    for k in range(4)], 1)                                      # [T,4]
n_reach = n_hit = n_blk_test = n_blk_hit = n_lane = 0
hit_all = torch.zeros(P, dtype=torch.bool, device=dev)
n_walk = int((idx < bmax.amax(1)[tile_of]).sum())
CH = 1 << 18
for a in range(0, P, CH):
    b = min(P, a + CH)
    tl = tile_of[a:b]; r = rec[g[a:b]]
    ux, uy, qxx, qxy, qyy, thr = r[:, 0], r[:, 1], r[:, 2], r[:, 3], r[:, 4], r[:, 11]
    tx0 = (tl % gx).float() * 16; ty0 = (tl // gx).float() * 16
    px = tx0[:, None, None] + torch.arange(16, device=dev).float()[None, None, :]
    py = ty0[:, None, None] + torch.arange(16, device=dev).float()[None, :, None]
    dx = ux[:, None, None] - px; dy = uy[:, None, None] - py
    pw = qxx[:, None, None] * dx * dx + qxy[:, None, None] * dx * dy + qyy[:, None, None] * dy * dy
    pix_hit = (idx[a:b, None, None] < cont_t[tl]) & (pw >= thr[:, None, None])    # [n,16,16]
    reach_e = torch.zeros(b - a, dtype=torch.bool, device=dev); hit_e = torch.zeros_like(reach_e)
    for k in range(4):
        rk = ((masks[a:b] >> k) & 1).bool() & (idx[a:b] < bmax[tl, k])
        hk = pix_hit[:, 8 * (k >> 1):8 * (k >> 1) + 8, 8 * (k & 1):8 * (k & 1) + 8].flatten(1)
        hk_any = hk.any(1) & rk
        n_blk_test += int(rk.sum()); n_blk_hit += int(hk_any.sum()); n_lane += int(hk[hk_any].sum())
        reach_e |= rk; hit_e |= hk_any
    n_reach += int(reach_e.sum()); n_hit += int(hit_e.sum())
    hit_all[a:b] = hit_e
print("P (list entries) %d; entries below the tile's largest contrib %d (%.3f)" % (P, n_walk, n_walk / P))
print("reach (todo) %d = %.3f of P; hit %d = %.3f of reach  -> a forward hit bit removes %.1f %% of the walked entries"
      % (n_reach, n_reach / P, n_hit, n_hit / n_reach, 100 * (1 - n_hit / n_reach)))
print("blocks tested %d (%.2f per reach entry), hit %d (%.2f per hit entry, %.3f of tested)"
      % (n_blk_test, n_blk_test / n_reach, n_blk_hit, n_blk_hit / max(n_hit, 1), n_blk_hit / n_blk_test))
print("lanes passing inside a hit block: %.1f of 64 (%.3f)" % (n_lane / n_blk_hit, n_lane / n_blk_hit / 64))

if "--time" in sys.argv:
    # VERDICT r3 task 4(a), priced from the backward side: hand k_draw_bwd the per-entry hit bits a forward pass
    # could leave behind (egs_probe_set_hit_bits) and time it against the same launch without them.
    import ctypes as C
    from easygaussiansplatting_amd import _lib
    from easygaussiansplatting_amd.function import GSFunction
    lib = _lib.load()
    words = (P + 31) // 32
    padded = torch.zeros(words * 32, dtype=torch.bool, device=dev); padded[:P] = hit_all
    w = (padded.view(words, 32).to(torch.int64) << torch.arange(32, device=dev)).sum(1)
    bits = (w & 0xFFFFFFFF).to(torch.int64)
    bits = torch.where(bits >= 2**31, bits - 2**32, bits).to(torch.int32).contiguous()
    GSFunction.mode = "fused"
    L = [t(sc.pws), t(sc.shs), t(sc.alphas).reshape(-1, 1), t(sc.scales), t(sc.rots)]
    for x in L:
        x.requires_grad_(True)
    us0 = torch.zeros((sc.n, 2), device=dev, requires_grad=True)
    dl = t(S.normal(1, 77, (3, H, W))) / (3 * H * W)

    # the probe is not in the product sources: git apply tools/lab/variants/draw_bwd_probes.patch, then
    # make FLAGS+=-DEGS_PROBE_HIT_BITS=1 (tools/lab/lab_r4d.sh); the symbol is bound here, not in _lib.SIGNATURES
    if not hasattr(lib, "egs_probe_set_hit_bits"):
        sys.exit("this libegs_hip.so was built without tools/lab/variants/draw_bwd_probes.patch")
    lib.egs_probe_set_hit_bits.restype = C.c_int; lib.egs_probe_set_hit_bits.argtypes = [C.c_void_p]
    if lib.egs_probe_set_hit_bits(None) != 0:
        sys.exit("this libegs_hip.so was built without -DEGS_PROBE_HIT_BITS=1 (tools/lab/lab_r4d.sh)")

    def run(probe, reps=40):
        lib.egs_probe_set_hit_bits(C.c_void_p(bits.data_ptr()) if probe else None)
        for r in range(reps + 10):
            if r == 10:
                lib.egs_prof_set_filter(b"k_draw_bwd"); lib.egs_prof_reset(); lib.egs_prof_enable(1)
            for x in L:
                x.grad = None
            us0.grad = None
            img, _ = GSFunction.apply(*L, us0, cam)
            img.backward(dl)
        torch.cuda.synchronize()
        lib.egs_prof_enable(0)
        need = lib.egs_prof_report(None, 0)
        buf = C.create_string_buffer(need + 16); lib.egs_prof_report(buf, need + 16)
        lib.egs_probe_set_hit_bits(None)
        row = [ln.split() for ln in buf.value.decode().splitlines() if ln.startswith("k_draw_bwd")][0]
        return float(row[2]) / int(row[1]) * 1e3, [x.grad.clone() for x in L]
    for rnd in range(3):
        t0, g0 = run(False)
        t1, g1 = run(True)
        worst = max(float((a - b).abs().max() / a.abs().max()) for a, b in zip(g0, g1))
        print("round %d: k_draw_bwd %.1f us without the bits, %.1f us with them (%.1f %%); gradients differ by %.1e of their maximum"
              % (rnd, t0, t1, 100 * (t1 / t0 - 1), worst))
