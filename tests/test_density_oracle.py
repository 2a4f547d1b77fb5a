"""Oracle (oracle/density_oracle.py) against fixture G8 -- the reference's own densification,
optimizer surgery and Adam run under CPU torch (tests/golden/make_golden_density.py)."""
import numpy as np

from oracle import density_oracle as D
from tests.conftest import load_golden

LRS = dict(zip(D.NAMES, (0.001, 0.001, 0.001 / 20, 0.05, 0.005, 0.001)))


def _state(g, prefix):
    return ({k: g[prefix + k] for k in D.NAMES}, {k: g[prefix + "m_" + k] for k in D.NAMES},
            {k: g[prefix + "v_" + k] for k in D.NAMES})


def test_adam_two_steps_match_torch():
    g = load_golden("g8_densify.npz")
    for k in D.NAMES:
        p = g["in_" + k]
        m = np.zeros_like(p); v = np.zeros_like(p)
        for s in range(2):
            p, m, v = D.adam_step(p, g["adam_grad%d_%s" % (s, k)], m, v, s + 1, LRS[k])
        np.testing.assert_allclose(p, g["pre_" + k], rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(m, g["pre_m_" + k], rtol=2e-6, atol=2e-11)
        np.testing.assert_allclose(v, g["pre_v_" + k], rtol=2e-6, atol=1e-15)


def test_density_statistics():
    g = load_golden("g8_densify.npz")
    acc = cnt = None
    for vw in range(3):
        acc, cnt = D.update_density_info(acc, cnt, g["view%d_dus" % vw], g["view%d_mask" % vw])
    np.testing.assert_array_equal(cnt, g["cunt"])
    np.testing.assert_allclose(acc, g["grad_accum"].reshape(-1), rtol=1e-6, atol=0)


def test_classification_is_exact():
    g = load_golden("g8_densify.npz")
    remain, clone, split = D.classify(g["pre_alphas_raw"], g["pre_scales_raw"], g["grad_accum"], g["cunt"],
                                      D.Thresholds(1.0))
    np.testing.assert_array_equal(remain, g["expect_remain"])
    np.testing.assert_array_equal(clone, g["expect_clone"])
    np.testing.assert_array_equal(split, g["expect_split"])
    assert len(clone) > 20 and len(split) > 20 and (~remain).sum() > 20


def test_densify_params_and_moments():
    g = load_golden("g8_densify.npz")
    p, m, v = _state(g, "pre_")
    p2, m2, v2, info = D.densify(p, m, v, g["grad_accum"], g["cunt"], g["unit_noise"], D.Thresholds(1.0))
    nk = int(info["remain"].sum())
    for k in D.NAMES:
        assert p2[k].shape == g["post_" + k].shape
        # surviving rows and moments are moved, not recomputed: bit-exact
        np.testing.assert_array_equal(p2[k][:nk], g["post_" + k][:nk])
        np.testing.assert_array_equal(m2[k], g["post_m_" + k])
        np.testing.assert_array_equal(v2[k], g["post_v_" + k])
        np.testing.assert_allclose(p2[k][nk:], g["post_" + k][nk:], rtol=3e-6, atol=3e-6)
        assert float(g["post_step_" + k]) == float(g["pre_step_" + k]) == 2.0


def test_split_keeps_parent_and_shrinks_child():
    g = load_golden("g8_densify.npz")
    p, m, v = _state(g, "pre_")
    p2, _, _, info = D.densify(p, m, v, g["grad_accum"], g["cunt"], g["unit_noise"], D.Thresholds(1.0))
    nk, nc = int(info["remain"].sum()), len(info["clone"])
    child = p2["scales_raw"][nk + nc:]
    parent = p["scales_raw"][info["split"]]
    np.testing.assert_allclose(child, parent + np.log(0.6), atol=2e-6)
    # the parent row is still there, untouched
    keep_pos = np.cumsum(info["remain"]) - 1
    np.testing.assert_array_equal(p2["scales_raw"][keep_pos[info["split"]]], parent)


def test_reset_alpha_and_following_adam_step():
    g = load_golden("g8_densify.npz")
    a = D.reset_alpha(g["post_alphas_raw"], D.Thresholds(1.0))
    np.testing.assert_allclose(a, g["reset_alphas_raw"], rtol=1e-6)
    assert not g["reset_m_alphas_raw"].any() and not g["reset_v_alphas_raw"].any()
    for k in D.NAMES:
        p, m, v = g["post_" + k], g["post_m_" + k], g["post_v_" + k]
        if k == "alphas_raw":
            p, m, v = g["reset_alphas_raw"], np.zeros_like(m), np.zeros_like(v)
        p, m, v = D.adam_step(p, g["adam_grad2_" + k], m, v, 3, LRS[k])
        np.testing.assert_allclose(p, g["final_" + k], rtol=3e-6, atol=1e-7)
        np.testing.assert_allclose(m, g["final_m_" + k], rtol=3e-6, atol=2e-11)


def test_lr_schedule():
    g = load_golden("g8_densify.npz")
    got = [D.expon_lr(int(s), 1e-4 * 2.5, 1e-6 * 2.5, 3000, delay_mult=0.01) for s in g["lr_steps"]]
    np.testing.assert_allclose(got, g["lr_values"], rtol=1e-12)
