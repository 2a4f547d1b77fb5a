"""``dist_views.ViewStreams`` on the host: lanes are detached aliases of the parameters with their own ``.grad``; the sum
over the lanes lands in the parameters' ``.grad``; a re-allocated parameter is re-aliased at ``begin()``."""
import pytest

torch = pytest.importorskip("torch")

from easygaussiansplatting_amd.dist_views import ViewStreams


def _loss(x, y, w):
    return (x * x).sum() * w + (y * 3.0).sum() * w


@pytest.mark.parametrize("n", [1, 2, 3])
def test_sum_over_lanes_equals_plain_accumulation(n):
    torch.manual_seed(0)
    a = torch.randn(7, 3, requires_grad=True)
    b = torch.randn(7, requires_grad=True)
    ws = [1.0, 2.0, 3.0, 0.5, 4.0]
    for w in ws:
        _loss(a, b, w).backward()
    ga, gb = a.grad.clone(), b.grad.clone()
    a.grad = None
    b.grad = None
    vs = ViewStreams([a, b], n)
    for rep in range(2):                      # the second step starts from empty accumulators
        a.grad = None
        b.grad = None
        vs.begin()
        for i, w in enumerate(ws):
            with vs.lane(i) as (x, y):
                assert x.data_ptr() == a.data_ptr() and (x is a) == (vs.lane_index(i) == 0)
                _loss(x, y, w).backward()
        vs.finish()
        assert torch.allclose(a.grad, ga) and torch.allclose(b.grad, gb)
        for lane in vs.leaves[1:]:
            assert all(q.grad is None for q in lane)


def test_lane_zero_without_views_adopts_the_other_accumulator():
    a = torch.ones(4, requires_grad=True)
    vs = ViewStreams([a], 2)
    vs.begin()
    with vs.lane(1) as (x,):                  # only the side lane renders
        (x * 5.0).sum().backward()
    vs.finish()
    assert torch.equal(a.grad, torch.full((4,), 5.0))


def test_reallocated_parameter_is_realiased():
    a = torch.ones(4, requires_grad=True)
    vs = ViewStreams([a], 2)
    with torch.no_grad():
        a.set_(torch.full((6,), 2.0))         # what densification does to the optimizer's tensors
    vs.begin()
    with vs.lane(1) as (x,):
        assert x.shape == (6,) and x.data_ptr() == a.data_ptr()
        (x * x).sum().backward()
    vs.finish()
    assert torch.equal(a.grad, torch.full((6,), 4.0))


def test_misuse_raises():
    a = torch.ones(3, requires_grad=True)
    vs = ViewStreams([a], 2)
    with pytest.raises(RuntimeError):
        vs.lane(0)
    with pytest.raises(RuntimeError):
        vs.finish()
    with pytest.raises(ValueError):
        ViewStreams([torch.ones(3)], 2)
    with pytest.raises(ValueError):
        ViewStreams([], 2)
