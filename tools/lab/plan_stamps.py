import sys, numpy as np, torch
sys.path.insert(0, ".")
from easygaussiansplatting_amd import fused, scene as S
from easygaussiansplatting_amd.function import Camera, GSFunction
sc = S.skewed_scene(reset_alpha=True)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
cam = Camera.from_scene(sc.cam)
P = dict(pws=dev(sc.pws), shs=dev(sc.shs), alphas=dev(sc.alphas).reshape(-1, 1), scales=dev(sc.scales), rots=dev(sc.rots))
for p in P.values(): p.requires_grad_(True)
dl = dev(S.normal(3, 21, (3, 1080, 1920)).astype(np.float32) / (3 * 1080 * 1920))
import gc
for it in range(4):
    us0 = torch.zeros((sc.n, 2), device="cuda", requires_grad=True)
    img, mask = GSFunction.apply(P["pws"], P["shs"], P["alphas"], P["scales"], P["rots"], us0, cam)
    st = [o for o in gc.get_objects() if isinstance(o, fused.FusedState) and o.seg is not None][-1]
    img.backward(dl)
    torch.cuda.synchronize()
    h = st.seg[:64 * 4].cpu().numpy().view(np.int32)
    f = h[16:22].astype(np.int64); b = h[32:38].astype(np.int64)
    print(it, "hdr", h[:8], "fwd stamps (10ns ticks, deltas)", np.diff(f), "bwd", np.diff(b))
    del st
