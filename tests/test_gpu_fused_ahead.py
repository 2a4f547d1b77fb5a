"""Fused forward with the draw stage enqueued ahead of the read-back of the patch count
(egs_splat_draw_rec_dev): same results as the synchronous path, safe on capacity / hint overflow."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(n, w, h, seed):
    from easygaussiansplatting_amd import scene as S
    from easygaussiansplatting_amd.function import Camera
    sc = S.small_scene(n, w, h, 12, seed=seed)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    return (dev(sc.pws), dev(sc.shs), dev(sc.alphas), dev(sc.scales), dev(sc.rots)), Camera.from_scene(sc.cam)


def _run(args, cam):
    from easygaussiansplatting_amd import fused
    img, mask, st = fused.forward(*args, cam)
    torch.cuda.synchronize()
    return [x.cpu().numpy() for x in (img, mask, st.ranges, st.gsid, st.contrib, st.final_tau, st.depths)]


@pytest.mark.parametrize("policy", ["gsplatcu", "forward_cpu"])
def test_ahead_equals_exact_and_survives_overflow(policy):
    from easygaussiansplatting_amd import fused
    from easygaussiansplatting_amd import gsplatcu as gsc
    gsc.set_policy(policy)
    try:
        args, cam = _scene(6000, 200, 120, 3)
        key = (6000, 200, 120)
        fused._patch_capacity.pop(key, None)
        ref = _run(args, cam)                               # no capacity yet: synchronous read-back
        assert fused._patch_capacity[key] > ref[3].shape[0] > 1000
        got = _run(args, cam)                               # enqueued ahead
        for a, b in zip(ref, got):
            np.testing.assert_array_equal(a, b)
        fused._patch_capacity[key] = 64                     # far too small: nothing out of bounds, draw redone
        got = _run(args, cam)
        for a, b in zip(ref, got):
            np.testing.assert_array_equal(a, b)
        assert fused._patch_capacity[key] > ref[3].shape[0]
        fused._patch_capacity[key] = ref[3].shape[0] - 7    # just too small
        got = _run(args, cam)
        for a, b in zip(ref, got):
            np.testing.assert_array_equal(a, b)
        gsc._key_bits_hint = 1                              # stale hint: detected from the returned max key
        got = _run(args, cam)
        for a, b in zip(ref, got):
            np.testing.assert_array_equal(a, b)
        assert 8 <= gsc._key_bits_hint <= 32
        s = torch.cuda.Stream()                             # and on a non-default stream
        with torch.cuda.stream(s):
            got = _run(args, cam)
        for a, b in zip(ref, got):
            np.testing.assert_array_equal(a, b)
        far = (args[0] + torch.tensor([0.0, 0.0, -1000.0], device="cuda"),) + args[1:]   # nothing visible
        out = _run(far, cam)
        assert out[3].shape[0] == 0 and not out[0].any() and not out[5].any()
    finally:
        gsc.set_policy("gsplatcu")
