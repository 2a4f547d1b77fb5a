import sys
sys.path.insert(0, ".")
import numpy as np, torch
from easygaussiansplatting_amd import scene as S, gsplatcu as gsc, fused
from easygaussiansplatting_amd.function import Camera, GSFunction, RenderOptions
dev = torch.device("cuda", 0)
sc = S.skewed_scene(reset_alpha=True)
cam = Camera.from_scene(sc.cam, dev)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
P = [t(sc.pws), t(sc.shs), t(sc.alphas).reshape(-1, 1).clone(), t(sc.scales), t(sc.rots)]
for p in P: p.requires_grad_(True)
us0 = torch.zeros((sc.n, 2), device=dev, requires_grad=True)
dl = torch.ones((3, sc.cam.height, sc.cam.width), device=dev) / 1e6
gsc.set_pair_states(True)
o = RenderOptions(mode="ops", ops_use_records=False)
orig = gsc._memo_sig
for i in range(4):
    for p in P: p.grad = None
    img, _ = GSFunction.apply(*P, us0, cam, o)
    e = gsc._pair_states.get((0, int(torch.cuda.current_stream().cuda_stream or 0)))
    print(i, "entry after splat:", None if e is None else (e["sig"][3], e["in_sig"][2]))
    img.backward(dl)
    torch.cuda.synchronize()
    print(i, gsc.last_splatB_info(), "keys", list(gsc._pair_states.keys()))
