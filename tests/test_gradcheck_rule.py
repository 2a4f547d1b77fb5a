"""Negative controls of the gradient acceptance rule (tests/gradcheck.py): it must REJECT what the round-3 rule
``|a-b| <= 2e-4 * max(1, |ref|max)`` accepted -- all-zeros, a 1 % scale error, one wrong large entry -- on gradients
of the size the parity tests produce (upstream gradient ~ 1/(H*W): entries of 1e-6 .. 1e-3)."""
import numpy as np
import pytest

from tests.gradcheck import assert_grad_close, grad_close, report


def _ref(seed=0, n=4000, scale=3e-7):
    rng = np.random.default_rng(seed)
    return rng.normal(size=(n, 3)) * scale * rng.lognormal(0, 1.5, size=(n, 1))


def test_accepts_fp32_rounding_and_counted_flips():
    ref = _ref()
    got = (ref * (1 + 3e-6 * np.random.default_rng(1).normal(size=ref.shape))).astype(np.float32)
    ok, r = grad_close(got, ref)
    assert ok and r["n_out"] == 0 and r["n_big"] > 100, r
    # two threshold-flip Gaussians: small absolute change, large relative change on entries just above 1 % of max
    sel = np.argsort(np.abs(ref[:, 0]))[::-1]
    rows = [i for i in sel if 0.012 * np.abs(ref).max() < abs(ref[i, 0]) < 0.02 * np.abs(ref).max()][:2]
    got2 = got.copy(); got2[rows, 0] *= 1.008
    assert not grad_close(got2, ref)[0]
    ok, r = grad_close(got2, ref, outliers=2)
    assert ok and r["n_out"] == 2, r


@pytest.mark.parametrize("what", ["zeros", "scaled_1.01", "scaled_0.99", "one_large_entry", "sign", "nan", "bulk_1e-3"])
def test_rejects(what):
    ref = _ref(2)
    got = ref.copy()
    if what == "zeros":
        got[:] = 0
    elif what.startswith("scaled"):
        got *= float(what.split("_")[1])
    elif what == "one_large_entry":
        i = np.unravel_index(np.argmax(np.abs(ref)), ref.shape)
        got[i] *= 1.01
    elif what == "sign":
        got = -got
    elif what == "nan":
        got[5, 1] = np.nan
    else:
        got *= 1 + 1e-3 * np.sign(np.random.default_rng(3).normal(size=ref.shape))
    # ... the round-3 rule took every one of them (except NaN)
    old = np.abs(got - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max())
    assert old or what == "nan"
    assert not grad_close(got, ref)[0], report(got, ref)
    with pytest.raises(AssertionError):
        assert_grad_close(got, ref, what)


def test_refuses_a_vacuous_reference_and_shape_mismatch():
    z = np.zeros((10, 3))
    assert not grad_close(z, z)[0]
    with pytest.raises(AssertionError):
        grad_close(np.zeros((10, 3)), np.zeros((10, 1, 3)))
