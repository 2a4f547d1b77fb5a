""""split ONE view into horizontal tile bands on side streams: band k's latency-bound launches
(emit, tile sort, ranges) run under band k-1's k_draw".  What could that buy?  Two measurements on the bench scene:

1. how much of a view's latency-bound part hides under ANOTHER stream's draw kernel at all: two forward-only renders
   (two ring cameras) one after the other on one stream against the same two on two streams;
2. what the banded chain itself would cost: the tile sort (the part of the chain behind the shared depth sort and scan
   that dominates it) on P, P/2 and P/4 patches -- the kernels are latency-bound, so a band's chain is NOT 1/B of the
   whole view's.

The ceiling of the banded forward is then  (chain(P) - chain(P/B))  minus what the concurrency costs the draw kernel."""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from easygaussiansplatting_amd import _lib, fused, scene as S          # noqa: E402
from easygaussiansplatting_amd.function import Camera                  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.load()
sc = S.big_scene(1_000_000, 1920, 1080, 48)
cams = [Camera.from_scene(c, dev) for c in S.ring_cameras(sc.cam, 8)[:2]]
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
P = [t(sc.pws), t(sc.shs), t(sc.alphas).reshape(-1, 1).clone(), t(sc.scales), t(sc.rots)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def seq():
    with torch.no_grad(), fused.deferred() as d:
        for c in cams:
            fused.forward(*P, c)
        d.commit()


def par():
    with torch.no_grad(), fused.deferred() as d:
        for s_, c in zip(streams, cams):
            s_.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s_):
                fused.forward(*P, c)
        for s_ in streams:
            torch.cuda.current_stream().wait_stream(s_)
        d.commit()


def timed(fn, n=60, warm=30):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


a, b = timed(seq), timed(par)
print("two forward-only renders: one stream %.4f ms (%.4f per view), two streams %.4f ms (%.4f per view): %.1f us hidden per view"
      % (a, a / 2, b, b / 2, (a - b) / 2 * 1e3))

# the tile sort alone at P, P/2, P/4 (13, 12, 11 tile bits: a band of 1/B of the tile rows)
for frac, bits in ((1, 13), (2, 12), (4, 11), (8, 10)):
    n = 3_640_000 // frac
    keys = torch.randint(0, 1 << bits, (n,), dtype=torch.int32, device=dev)
    vals = torch.arange(n, dtype=torch.int32, device=dev)
    ka, va = torch.empty_like(keys), torch.empty_like(vals)
    ws = torch.empty(lib.egs_sort_pairs_ws_bytes(n), dtype=torch.uint8, device=dev)
    flag = C.c_int(0)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda x: C.c_void_p(x.data_ptr())

    def sort():
        _lib.check(lib.egs_sort_pairs(n, p(keys), p(vals), p(ka), p(va), 0, bits, p(ws), ws.numel(), C.byref(flag), st))
    ms = timed(sort, 200, 50)
    print("tile sort of %8d patches on %2d bits (%d passes + the 1-launch fill of the superblock sums): %.1f us"
          % (n, bits, (bits + 7) // 8, ms * 1e3))
