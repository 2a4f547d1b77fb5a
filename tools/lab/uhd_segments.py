"""Ad hoc: the segment path at 3840 x 2160 (32 400 tiles: beyond TILE_ORDER_MAX_T, 15 tile bits) on scene.skewed_scene right
after reset_alpha seen through a camera of twice the focal length, against the unsplit kernels (tests/test_gpu_segments.py's
own comparison)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from easygaussiansplatting_amd import fused, gsplatcu, scene as S      # noqa: E402
from easygaussiansplatting_amd.function import Camera                  # noqa: E402
from tests.test_gpu_segments import compare, dev, run                  # noqa: E402

gsplatcu.set_policy("gsplatcu")
W, H = 3840, 2160
sc = S.skewed_scene(reset_alpha=True)
sc.cam = S.Camera(W, H, 2 * sc.cam.fx, 2 * sc.cam.fy, W / 2.0, H / 2.0, sc.cam.Rcw, sc.cam.tcw)
dl = dev(S.normal(3, 22, (3, H, W)).astype(np.float32) / (3 * H * W))
fused.SEGMENTS = "0"
ref = run(fused, sc, Camera.from_scene(sc.cam), dl)
fused.SEGMENTS = "auto"
got = run(fused, sc, Camera.from_scene(sc.cam), dl, 2)
lens = ref["ranges"][:, 1] - ref["ranges"][:, 0]
print("tiles", lens.size, "longest list", lens.max(), "longest walk", ref["contrib"].max(), "segment path", got["seg"])
compare(got, ref, "uhd_skewed_reset", flips=256)
print("OK")
