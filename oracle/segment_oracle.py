"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of the SEGMENT form of the per-tile blend, on top of gs_oracle.

The reference blends a tile's list front to back with one thread per pixel (gsplatcu/kernel.cu:152-271) and walks it
back to front for the gradients (kernel.cu:809-950); `gs_oracle.draw` / `draw_backward` restate exactly that.  The
product splits LONG lists into segments of L entries that are blended by different waves (DESIGN 3.5,
csrc/egs_raster.hip k_draw_seg / k_draw_bwd<SEG>).  This module states that decomposition in float64 NumPy, step by
step as the kernels perform it, so that `tests/test_segment_oracle.py` can show ON THE CPU that it is the same function
as the reference's loop -- including the early stop (kernel.cu:256-260), which is the part that is NOT associative:

  forward   1. every segment s is blended from tau = 1 ("local frame"): (C_s, tau_s, last contributor); a pixel stops
               only when its LOCAL tau < tau_stop
            2. T_s = tau_0 ... tau_(s-1), multiplied in that order
            3. fix: a pixel with T_s >= tau_stop > T_s tau_s finishes INSIDE segment s: it is blended again from
               tau = T_s with the reference's stop rule, its state put back into the local frame, its last contributor
               marked "finished here"
            4. compose in order: colour += T C_s, T *= tau_s while the pixel is alive
            5. what the backward pass needs at the END of segment s: the transmittance there and G_s, the colour of
               everything behind it seen from there: G_last = 0, G_(s-1) = C_s + tau_s G_s
  backward  segment s alone: a pixel whose last contributor lies behind the segment starts from (T_end(s), G_s) -- the
            reference's loop starts every pixel from (final_tau, 0) at its last contributor -- and walks [s L, (s+1) L).

Nothing in the product imports this file."""
import numpy as np

from . import gs_oracle as O


def _tile_pixels(t, gx, width, height, dtype):
    ty, tx = divmod(t, gx)
    y0, x0 = ty * O.TILE, tx * O.TILE
    hh, ww = min(O.TILE, height - y0), min(O.TILE, width - x0)
    py, px = np.meshgrid(np.arange(y0, y0 + hh, dtype=dtype), np.arange(x0, x0 + ww, dtype=dtype), indexing="ij")
    return y0, x0, hh, ww, py, px


def _walk(entries, r0, tau0, alive0, gsid, us, cinv2ds, alphas, colors, px, py, policy, dtype):
    """the reference's blend loop (gs_oracle.draw's body, kernel.cu:228-262) over the list entries `entries` (0-based
    positions in the tile's list) from transmittance tau0, for the pixels `alive0` -> (colour, tau, last contributor)"""
    tau = np.array(tau0, dtype)
    col = np.zeros((3,) + tau.shape, dtype)
    cont = np.zeros(tau.shape, np.int32)
    done = ~alive0
    for e in entries:
        if done.all():
            break
        g = int(gsid[r0 + e])
        ap, _, _, _ = O._alpha_prime(alphas[g], cinv2ds[g], us[g], px, py, policy, dtype)
        act = ~done & ~(ap < dtype(policy.alpha_skip))
        w = np.where(act, tau * ap, 0)
        col += w[None] * colors[g][:, None, None]
        cont = np.where(act, e + 1, cont)
        tau = np.where(act, tau * (1 - ap), tau)
        done |= act & (tau < dtype(policy.tau_stop))
    return col, tau, cont


def draw_segments(width, height, ranges, gsid, us, cinv2ds, alphas, colors, L, policy=O.POLICY_G, tiles=None,
                  dtype=np.float64):
    """-> image[3,H,W], contrib[H,W], final_tau[H,W], states {tile: (G[S,3,h,w], T_end[S,h,w])}: the five steps of the
    module docstring for every tile (every tile is "split", whatever its length)."""
    assert policy.footprint != O.FOOT_BOX and policy.alpha_skip > 0 and policy.tau_stop > 0
    gx, gy = O.tile_grid(width, height)
    us = np.asarray(us, dtype); cinv2ds = np.asarray(cinv2ds, dtype)
    alphas = np.asarray(alphas, dtype).reshape(-1); colors = np.asarray(colors, dtype)
    stop = dtype(policy.tau_stop)
    image = np.zeros((3, height, width), dtype)
    contrib = np.zeros((height, width), np.int32)
    final_tau = np.zeros((height, width), dtype)
    states = {}
    for t in (range(gx * gy) if tiles is None else tiles):
        r0, r1 = int(ranges[t, 0]), int(ranges[t, 1])
        n = r1 - r0
        if n == 0:
            continue
        y0, x0, hh, ww, py, px = _tile_pixels(t, gx, width, height, dtype)
        S = (n + L - 1) // L
        every = np.ones((hh, ww), bool)
        walk = lambda s, tau0, alive: _walk(range(s * L, min(n, (s + 1) * L)), r0, tau0, alive, gsid, us, cinv2ds,
                                            alphas, colors, px, py, policy, dtype)
        # 1. local frames
        C = np.zeros((S, 3, hh, ww), dtype); tl = np.ones((S, hh, ww), dtype); lc = np.zeros((S, hh, ww), np.int32)
        for s in range(S):
            C[s], tl[s], lc[s] = walk(s, np.ones((hh, ww), dtype), every)
        # 2. the transmittance in front of every segment, and 3. the pixels that finish inside one
        T = np.ones((hh, ww), dtype)
        for s in range(S):
            if s > 0:
                fin = (T >= stop) & (T * tl[s] < stop)
                if fin.any():
                    c_abs, t_abs, k = walk(s, np.where(fin, T, 1), fin)
                    C[s] = np.where(fin[None], c_abs / np.where(fin, T, 1)[None], C[s])
                    tl[s] = np.where(fin, t_abs / np.where(fin, T, 1), tl[s])
                    lc[s] = np.where(fin, np.where(t_abs < stop, -k, k), lc[s])
            T = T * tl[s]
        # 4. composition
        Tf = np.ones((hh, ww), dtype)           # negative: finished, |Tf| the final transmittance
        col = np.zeros((3, hh, ww), dtype)
        cont = np.zeros((hh, ww), np.int32)
        T_end = np.zeros((S, hh, ww), dtype)
        sdone = 0
        for s in range(S):
            alive = Tf >= stop
            if not alive.any():
                break
            sdone = s + 1
            col += np.where(alive, Tf, 0)[None] * C[s]
            tn = Tf * tl[s]
            finished = alive & ((lc[s] < 0) | (tn < stop))
            cont = np.where(alive & (lc[s] != 0), np.abs(lc[s]), cont)
            Tf = np.where(alive, np.where(finished, -np.maximum(tn, 1e-300), tn), Tf)
            T_end[s] = np.where(alive, np.abs(Tf), T_end[s])
        image[:, y0:y0 + hh, x0:x0 + ww] = col
        contrib[y0:y0 + hh, x0:x0 + ww] = cont
        final_tau[y0:y0 + hh, x0:x0 + ww] = np.abs(Tf)
        # 5. the states at the segments' ends
        sstar = np.where(Tf < stop, np.maximum(cont - 1, 0) // L, sdone - 1)
        G = np.zeros((S, 3, hh, ww), dtype)
        acc = np.zeros((3, hh, ww), dtype)
        for s in range(sdone - 1, -1, -1):
            G[s] = acc
            acc = np.where((s <= sstar)[None], C[s] + tl[s][None] * acc, acc)
        states[t] = (G, T_end)
    return image, contrib, final_tau, states


def draw_backward_segments(width, height, ranges, gsid, us, cinv2ds, alphas, colors, contrib, final_tau, dloss_dgammas,
                           states, L, policy=O.POLICY_G, tiles=None, dtype=np.float64):
    """-> dloss_dus[N,2], dloss_dcinv2ds[N,3], dloss_dalphas[N], dloss_dcolors[N,3]: every segment of every tile walked
    on its own from the state at its end (gs_oracle.draw_backward's loop body, kernel.cu:880-948)."""
    gx, gy = O.tile_grid(width, height)
    us = np.asarray(us, dtype); cinv2ds = np.asarray(cinv2ds, dtype)
    alphas = np.asarray(alphas, dtype).reshape(-1); colors = np.asarray(colors, dtype)
    dLdg = np.asarray(dloss_dgammas, dtype)
    n_g = us.shape[0]
    dus = np.zeros((n_g, 2), dtype); dcinv = np.zeros((n_g, 3), dtype)
    dalpha = np.zeros(n_g, dtype); dcolor = np.zeros((n_g, 3), dtype)
    for t in (range(gx * gy) if tiles is None else tiles):
        r0, r1 = int(ranges[t, 0]), int(ranges[t, 1])
        n = r1 - r0
        if n == 0:
            continue
        y0, x0, hh, ww, py, px = _tile_pixels(t, gx, width, height, dtype)
        cont_all = np.asarray(contrib[y0:y0 + hh, x0:x0 + ww])
        dl = dLdg[:, y0:y0 + hh, x0:x0 + ww]
        G, T_end = states[t]
        for s in range((n + L - 1) // L):          # any order: the segments do not depend on each other
            lo, hi = s * L, min(n, (s + 1) * L)
            behind = cont_all > hi                 # contributors behind this segment: start from the state at its end
            tau = np.where(behind, T_end[s], np.array(final_tau[y0:y0 + hh, x0:x0 + ww], dtype))
            gcl = np.where(behind[None], G[s], 0)
            cont = np.where(behind, hi, np.where(cont_all > lo, cont_all, 0))
            if cont.max() <= lo:
                continue
            for e in range(int(cont.max()) - 1, lo - 1, -1):
                g = int(gsid[r0 + e])
                ap, gg, dx, dy = O._alpha_prime(alphas[g], cinv2ds[g], us[g], px, py, policy, dtype)
                act = (e < cont) & ~(ap < dtype(policy.alpha_skip))
                if not act.any():
                    continue
                with np.errstate(all="ignore"):
                    tau_n = np.where(act, tau / (1 - ap), tau)
                c = colors[g][:, None, None]
                dl_dap = np.where(act, (dl * (tau_n[None] * (c - gcl))).sum(0), 0)
                dalpha[g] += (dl_dap * gg).sum()
                dcolor[g] += (np.where(act, ap * tau_n, 0)[None] * dl).sum((1, 2))
                ci = cinv2ds[g]
                dus[g, 0] += (dl_dap * (-ci[0] * dx - ci[1] * dy) * ap).sum()
                dus[g, 1] += (dl_dap * (-ci[1] * dx - ci[2] * dy) * ap).sum()
                dcinv[g, 0] += (dl_dap * (-0.5 * ap * dx * dx)).sum()
                dcinv[g, 1] += (dl_dap * (-1.0 * ap * dx * dy)).sum()
                dcinv[g, 2] += (dl_dap * (-0.5 * ap * dy * dy)).sum()
                gcl = np.where(act[None], ap[None] * c + (1 - ap)[None] * gcl, gcl)
                tau = tau_n
    return dus, dcinv, dalpha, dcolor
