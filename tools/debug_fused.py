import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easygaussiansplatting_amd import scene as S, fused, gsplatcu as gsc
from easygaussiansplatting_amd.function import Camera, GSFunction
dev = torch.device("cuda", 0)
mode = sys.argv[1] if len(sys.argv) > 1 else "fused"
for n in (10_000, 200_000, 1_000_000):
    sc = S.big_scene(n)
    cam = Camera.from_scene(sc.cam, dev)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
    P = dict(pws=t(sc.pws), shs=t(sc.shs), alphas=t(sc.alphas).reshape(-1, 1), scales=t(sc.scales), rots=t(sc.rots))
    dl = torch.ones((3, 1080, 1920), device=dev) / 1e6
    if mode == "fused":
        img, mask, st = fused.forward(P["pws"], P["shs"], P["alphas"], P["scales"], P["rots"], cam)
        torch.cuda.synchronize(); print(n, "fwd ok", st.gsid.shape[0], float(img.mean()), flush=True)
        g = fused.backward(P["pws"], P["shs"], P["alphas"], P["scales"], P["rots"], cam, st, dl)
        torch.cuda.synchronize(); print(n, "bwd ok", [float(x.abs().max()) for x in g], flush=True)
    else:
        GSFunction.mode = mode
        for p in P.values(): p.requires_grad_(True)
        us0 = torch.zeros((n, 2), device=dev, requires_grad=True)
        img, mask = GSFunction.apply(P["pws"], P["shs"], P["alphas"], P["scales"], P["rots"], us0, cam)
        torch.cuda.synchronize(); print(n, "fwd ok", float(img.mean()), flush=True)
        img.backward(dl)
        torch.cuda.synchronize(); print(n, "bwd ok", float(P["pws"].grad.abs().max()), flush=True)
