"""Memory safety without a sanitizer (GPU AddressSanitizer is not available on this pool): every device buffer the host
layer allocates for the kernels -- records, lists, workspaces, segment states, outputs (``torch.empty`` in fused.py and
gsplatcu.py) -- gets a 4-KB guard band of 0xA5 bytes on either side; after forward + backward through every path (fused,
the seven ops with the handle and as the public pair, a ragged 17 x 33 image, dense long lists on 64-entry segments with
every kind of work item, an enqueue-ahead capacity that overflows) every band is untouched.  (The full sweep, incl. the
1080p scenes: tools/lab/guard_bands.py -> profiles/r6_guard_bands.txt.)"""
import ctypes as C

import numpy as np
import pytest

from easygaussiansplatting_amd import scene as S

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
GUARD = 4096


class _Proxy:
    def __init__(self, bands):
        self._bands = bands

    def __getattr__(self, name):
        return getattr(torch, name)

    def empty(self, *size, dtype=None, device=None, **kw):
        if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
            size = tuple(size[0])
        size = tuple(int(s) for s in size)
        dt = dtype or torch.float32
        dv = torch.device(device) if device is not None else torch.device("cpu")
        if dv.type != "cuda" or kw:
            return torch.empty(size, dtype=dt, device=dv, **kw)
        nbytes = int(np.prod(size, dtype=np.int64)) * torch.empty((), dtype=dt).element_size()
        base = torch.empty(nbytes + (-nbytes) % 256 + 2 * GUARD, dtype=torch.uint8, device=dv)
        base[:GUARD] = 0xA5
        base[GUARD + nbytes:] = 0xA5
        self._bands.append((base, nbytes))
        return base[GUARD:GUARD + nbytes].view(dt).reshape(size)


def test_no_kernel_writes_outside_its_buffers():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from easygaussiansplatting_amd import _lib, fused, gsplatcu as gsc
    from easygaussiansplatting_amd.function import Camera, GSFunction, RenderOptions
    gsc.set_policy("gsplatcu")
    lib = _lib.load()
    bands = []
    before = (C.c_int * 2)()
    _lib.check(lib.egs_seg_config(0, 0, before))
    keep = (fused.SEGMENTS, fused.SEG_SPECULATE, GSFunction.mode)
    fused.torch = gsc.torch = _Proxy(bands)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()

    def step(sc, cam, opts=None, reps=2):
        P = [dev(sc.pws), dev(sc.shs), dev(sc.alphas).reshape(-1, 1), dev(sc.scales), dev(sc.rots)]
        for p in P:
            p.requires_grad_(True)
        us0 = torch.zeros((sc.n, 2), device="cuda", requires_grad=True)
        dl = dev(S.normal(3, 5, (3, cam.height, cam.width)).astype(np.float32) / (3 * cam.height * cam.width))
        for _ in range(reps):
            for p in P:
                p.grad = None
            img, _ = GSFunction.apply(*P, us0, cam, opts) if opts is not None else GSFunction.apply(*P, us0, cam)
            img.backward(dl)
        assert all(torch.isfinite(p.grad).all() for p in P)

    def check(tag):
        torch.cuda.synchronize()
        assert len(bands) > 20, tag
        for base, nbytes in bands:
            assert bool((base[:GUARD] == 0xA5).all()) and bool((base[GUARD + nbytes:] == 0xA5).all()), (tag, nbytes)
        del bands[:]

    try:
        GSFunction.mode = "fused"
        ops = (None, RenderOptions(mode="ops"), RenderOptions(mode="ops", ops_use_records=False))
        for (n, W, H, K) in ((5_000, 17, 33, 3), (20_000, 200, 120, 48)):
            sc = S.small_scene(n, W, H, K, seed=3)
            for o in ops:
                step(sc, Camera.from_scene(sc.cam), o)
            check("%d %dx%d" % (n, W, H))
        sc = S.small_scene(60_000, 320, 240, 12, seed=5)
        sc.scales[:] = sc.scales * 2.2
        _lib.check(lib.egs_seg_config(64, 64, None))
        for reset in (False, True):
            if reset:
                sc.alphas[:] = np.minimum(sc.alphas, 0.01)
            for spec in ("0", "1"):
                fused.SEGMENTS, fused.SEG_SPECULATE = "1", spec
                for o in ops:
                    step(sc, Camera.from_scene(sc.cam), o)
            check("segments of 64, %s" % ("reset" if reset else "opaque"))
        fused.SEGMENTS, fused.SEG_SPECULATE = keep[0], keep[1]
        sc = S.small_scene(20_000, 200, 120, 48, seed=3)
        cam = Camera.from_scene(sc.cam)
        step(sc, cam)
        for cap in (64, 1000, 30_000):      # the draw stage on truncated lists, then again
            fused._ctx(torch.device("cuda", torch.cuda.current_device())).capacity[(sc.n, 200, 120)] = cap
            step(sc, cam, reps=1)
        check("overflowing enqueue-ahead capacity")
    finally:
        fused.torch = gsc.torch = torch
        fused.SEGMENTS, fused.SEG_SPECULATE, GSFunction.mode = keep
        _lib.check(lib.egs_seg_config(before[0], before[1], None))
