"""tests/oracle_parallel.py == the serial oracle (same arithmetic, tiles dealt to worker processes)."""
import numpy as np

from easygaussiansplatting_amd import scene as S
from oracle import gs_oracle as O
from tests.oracle_parallel import draw_backward_tiles


def test_parallel_backward_equals_serial():
    sc = S.small_scene(1500, 160, 96, 3, seed=12)
    cam = sc.cam
    st = O.forward_pipeline((sc.pws, sc.rots, sc.scales, sc.alphas, sc.shs), cam, O.POLICY_G)
    dl = S.normal(3, 5, (3, cam.height, cam.width)) / (cam.height * cam.width)
    a64 = sc.alphas.astype(np.float64)
    near_s = np.zeros(sc.n, bool)
    want = O.draw_backward(cam.width, cam.height, st["ranges"], st["gsid"], st["us"], st["cinv2ds"], a64, st["colors"],
                           st["contrib"], st["final_tau"], dl, None, O.POLICY_G, near_out=near_s)
    for procs in (1, 3):
        got = draw_backward_tiles(cam.width, cam.height, st["ranges"], st["gsid"], st["us"], st["cinv2ds"], a64,
                                  st["colors"], st["contrib"], st["final_tau"], dl, procs=procs)
        for x, y in zip(got[:4], want):
            assert np.abs(x - y).max() <= 1e-15 * max(1.0, np.abs(y).max()) + 1e-22      # summation order only
        assert np.array_equal(got[4], near_s)
    sub = np.array([0, 7, 13, 59])
    got = draw_backward_tiles(cam.width, cam.height, st["ranges"], st["gsid"], st["us"], st["cinv2ds"], a64,
                              st["colors"], st["contrib"], st["final_tau"], dl, tiles=sub, procs=2)
    want = O.draw_backward(cam.width, cam.height, st["ranges"], st["gsid"], st["us"], st["cinv2ds"], a64, st["colors"],
                           st["contrib"], st["final_tau"], dl, None, O.POLICY_G, tiles=sub)
    for x, y in zip(got[:4], want):
        assert np.abs(x - y).max() <= 1e-15 * max(1.0, np.abs(y).max()) + 1e-22
