"""Does a hipGraph buy anything on the launch-bound binning chain?  (The host already enqueues a whole step ahead: the 16
kernels of a step sit in the queue back to back, profiles/r6_step_timeline.txt shows 0.0 us idle -- what a graph can
still remove is what the command processor spends per dependent dispatch.)  One forward-only render of the bench scene
captured with torch.cuda.CUDAGraph (stream capture of the library's launches) and replayed, against the same render
enqueued the ordinary way."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from easygaussiansplatting_amd import fused, scene as S          # noqa: E402
from easygaussiansplatting_amd.function import Camera            # noqa: E402

dev = torch.device("cuda", 0)
sc = S.big_scene(1_000_000, 1920, 1080, 48)
cam = Camera.from_scene(sc.cam, dev)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
P = [t(sc.pws), t(sc.shs), t(sc.alphas).reshape(-1, 1).clone(), t(sc.scales), t(sc.rots)]
fused.SEGMENTS = "0"


def render():
    with torch.no_grad(), fused.deferred() as d:
        out = fused.forward(*P, cam)
        d.commit()
    return out


def timed(fn, n=200, warm=50):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(5):
    render()
torch.cuda.synchronize()
print("ordinary enqueue: %.4f ms per forward render" % timed(render))
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
try:
    with torch.cuda.stream(side):
        with torch.no_grad():
            for _ in range(3):
                fused.forward(*P, cam)            # (warm-up on the capture stream: its own walk word, order buffer)
        torch.cuda.synchronize()
        g.capture_begin()
        with torch.no_grad():
            tls_prev = getattr(fused._tls, "deferred", False)
            fused._tls.deferred = True            # (no host wait inside the capture: validation is not part of the graph)
            img, mask, st = fused.forward(*P, cam)
            fused._tls.deferred = tls_prev
        g.capture_end()
    # (lab only: the captured render's read-back ticket would be validated against a slot its kernels never wrote during
    # the capture; the replays write the slot, nobody reads it)
    ctx = fused._ctx(dev)
    with ctx.lock:
        if st.ticket in ctx.pending:
            ctx.pending.remove(st.ticket)
    st.ticket = None
    torch.cuda.synchronize()
    ref = render()[0]
    g.replay()
    torch.cuda.synchronize()
    print("captured; image equal to the ordinary render:", bool(torch.equal(img, ref)))
    print("graph replay:     %.4f ms per forward render" % timed(g.replay))
except Exception as e:                            # noqa: BLE001
    print("capture failed:", type(e).__name__, str(e)[:400])
