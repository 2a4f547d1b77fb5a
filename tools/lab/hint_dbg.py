import sys
sys.path.insert(0, ".")
import ctypes as C
import numpy as np, torch
from easygaussiansplatting_amd import scene as S, fused, _lib, gsplatcu
from easygaussiansplatting_amd.function import Camera
gsplatcu.set_policy("gsplatcu")
lib = _lib.load()
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
host = lambda t: t.detach().cpu().numpy()
W, H = 320, 240
sc = S.small_scene(60_011, W, H, 3, seed=7)
sc.scales[:] = sc.scales * 2.2
lo = np.minimum(sc.alphas, 0.01).astype(np.float32); hi = sc.alphas.astype(np.float32)
cam_a = Camera.from_scene(sc.cam); cam_b = Camera.from_scene(S.ring_cameras(sc.cam, 8)[3])
key = (sc.n, W, H); d = torch.device("cuda", 0)
_lib.check(lib.egs_seg_config(64, 500, None))
P = [dev(sc.pws), dev(sc.shs), None, dev(sc.scales), dev(sc.rots)]
for mode in ("auto", "0", "1"):
    fused.SEGMENTS = mode
    print("SEGMENTS", mode)
    for i, (alphas, cam, nm) in enumerate(((lo, cam_a, "lo a"), (hi, cam_b, "hi b"), (hi, cam_a, "hi a"), (lo, cam_b, "lo b"), (lo, cam_a, "lo a"), (hi, cam_b, "hi b"), (hi, cam_b, "hi b"), (hi, cam_b, "hi b"))):
        P[2] = dev(alphas).reshape(-1, 1)
        before = fused.seg_hint(d, key)
        with torch.no_grad():
            _, _, st = fused.forward(*P, cam, need_grad=False)
        torch.cuda.synchronize()
        rg, ct = host(st.ranges), host(st.contrib)
        print(i, nm, "own pair", (int((rg[:, 1] - rg[:, 0]).max()), int(ct.max())), "seg", st.seg is not None, "hint before", before, "after", fused.seg_hint(d, key))
