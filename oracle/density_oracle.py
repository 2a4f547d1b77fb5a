"""CPU oracle for the densification / optimizer-surgery row.

TEST INFRASTRUCTURE ONLY: imported by ``tests/`` (and nothing in the product
path).  A NumPy restatement, in float32 like the reference's torch code, of

* ``GSModel.update_density_info``      gsplat/gsmodel.py:214-230
* ``GSModel.update_gaussian_density``  gsplat/gsmodel.py:232-317
  (+ ``prune_params`` 151-166, ``update_params`` 132-148)
* ``GSModel.reset_alpha``              gsplat/gsmodel.py:319-330
* ``rotate_vector_by_quaternion``      gsplat/utils.py:46-54
* ``torch.optim.Adam`` as configured by train.py:32 (lr per group, betas (0.9, 0.999),
  eps 1e-15, no weight decay, no amsgrad) -- the published Adam update with torch's
  operation order.

Parity pin: fixture ``tests/golden/g8_densify.npz`` produced by running the
reference's own functions under CPU torch (tests/golden/make_golden_density.py).

Reference behaviours that are easy to miss and are restated on purpose:
* the first ``update_density_info`` call stores the gradient norm of EVERY Gaussian and
  the mask as counts; later calls add only where the mask is set (l.222-228);
* pruning happens first; clone/split are decided on the survivors (l.234-255);
* the split keeps the ORIGINAL Gaussian unchanged (the ``scales[...] *= 0.6`` at l.279
  writes into a temporary) and appends ONE new Gaussian with 0.6 x scale at
  ``pw + R(q) . N(0, scale)``; clones are appended unchanged;
* appended rows store ``logit(sigmoid(alpha_raw))``, ``log(exp(scale_raw))`` and the
  NORMALISED quaternion (l.283-288), their Adam moments are zero and the group's
  ``step`` is kept (l.132-148);
* ``grads = grad_accum / cunt`` with 0/0 -> 0 (l.241-242).
"""
import numpy as np

F = np.float32
NAMES = ("pws", "low_shs", "high_shs", "alphas_raw", "scales_raw", "rots_raw")
WIDTHS = (3, 3, 45, 1, 3, 4)


def logit(x):
    return float(np.log(x / (1 - x)))


class Thresholds:
    """GSModel.__init__ (gsmodel.py:170-179)."""

    def __init__(self, scene_size=1.0):
        self.grad = 4e-7
        self.scale = 0.01 * scene_size
        self.alpha = 0.005
        self.big = 0.1 * scene_size
        self.reset_alpha = 0.01


def update_density_info(grad_accum, cunt, dus, mask):
    """gsmodel.py:214-230.  Pass ``grad_accum=None`` for the first view.  Returns the new pair."""
    g = np.sqrt((dus.astype(F) ** 2).sum(-1, dtype=F)).astype(F)
    if grad_accum is None:
        return g.copy(), mask.astype(np.int32)
    grad_accum = grad_accum.copy()
    cunt = cunt + mask.astype(np.int32)
    grad_accum[mask] += g[mask]
    return grad_accum, cunt


def classify(alphas_raw, scales_raw, grad_accum, cunt, th):
    """gsmodel.py:234-255 -> (remain mask [N], clone idx, split idx) in ORIGINAL indices."""
    a = alphas_raw.reshape(-1).astype(F)
    smax_raw = scales_raw.astype(F).max(axis=1)
    prune = (a < F(logit(th.alpha))) | (smax_raw > F(np.log(th.big)))
    remain = ~prune
    with np.errstate(divide="ignore", invalid="ignore"):
        grads = grad_accum.reshape(-1).astype(F) / cunt.astype(F)
    grads[np.isnan(grads)] = 0
    by_grad = grads >= F(th.grad)
    by_scale = np.exp(smax_raw) <= F(th.scale)          # max(exp(x)) == exp(max(x))
    idx = np.arange(a.shape[0])
    return remain, idx[remain & by_grad & by_scale], idx[remain & by_grad & ~by_scale]


def rotate_vector_by_quaternion(q, v):
    """utils.py:46-54 (q = (w, x, y, z), normalised inside)."""
    q = q.astype(F)
    q = q / np.maximum(np.linalg.norm(q, axis=1, keepdims=True), F(1e-12))
    u, s = q[:, 1:], q[:, :1]
    v = v.astype(F)
    return (F(2) * u * (u * v).sum(1, keepdims=True) + v * (s * s - (u * u).sum(1, keepdims=True))
            + F(2) * s * np.cross(u, v)).astype(F)


def densify(params, m, v, grad_accum, cunt, unit_noise, th):
    """gsmodel.py:232-317.  ``params``/``m``/``v``: dicts name -> [N, w] float32 (Adam moments may be
    None = optimizer has not stepped yet).  ``unit_noise`` [N,3]: unit normals indexed by ORIGINAL
    Gaussian index (the reference draws torch.normal(0, scale) = scale * unit).  Returns
    (params', m', v', info)."""
    remain, clone, split = classify(params["alphas_raw"], params["scales_raw"], grad_accum, cunt, th)
    sig = lambda x: (F(1) / (F(1) + np.exp(-x.astype(F)))).astype(F)
    norm = lambda q: (q / np.maximum(np.linalg.norm(q.astype(F), axis=1, keepdims=True), F(1e-12))).astype(F)
    new = {}
    for name in NAMES:
        new[name] = [params[name][remain]]
    for sel, is_split in ((clone, False), (split, True)):
        pws = params["pws"][sel].astype(F)
        alphas = sig(params["alphas_raw"][sel])
        scales = np.exp(params["scales_raw"][sel].astype(F)).astype(F)
        rots = norm(params["rots_raw"][sel])
        if is_split:
            pws = pws + rotate_vector_by_quaternion(rots, scales * unit_noise[sel].astype(F))
            scales = scales * F(0.6)
        new["pws"].append(pws.astype(F))
        new["low_shs"].append(params["low_shs"][sel])
        new["high_shs"].append(params["high_shs"][sel])
        new["alphas_raw"].append(np.log(alphas / (F(1) - alphas)).astype(F))
        new["scales_raw"].append(np.log(scales).astype(F))
        new["rots_raw"].append(rots)
    out = {k: np.concatenate(vv, axis=0).astype(F) for k, vv in new.items()}
    n_new = len(clone) + len(split)

    def surg(st):
        if st is None:
            return None
        return {k: np.concatenate([st[k][remain], np.zeros((n_new,) + st[k].shape[1:], F)], axis=0) for k in NAMES}

    info = {"remain": remain, "clone": clone, "split": split}
    return out, surg(m), surg(v), info


def reset_alpha(alphas_raw, th):
    """gsmodel.py:319-330: clamp from above to logit(0.01); the caller zeroes both Adam moments."""
    r = F(logit(th.reset_alpha))
    return np.where(alphas_raw > r, r, alphas_raw).astype(F)


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-15):
    """One torch.optim.Adam update (train.py:32 configuration); ``step`` is the 1-based count AFTER
    the increment.  Returns (p', m', v')."""
    p, g, m, v = (x.astype(F) for x in (p, g, m, v))
    m = (m + (g - m) * F(1 - beta1)).astype(F)          # torch: exp_avg.lerp_(grad, 1 - beta1)
    v = (v * F(beta2) + g * g * F(1 - beta2)).astype(F)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (np.sqrt(v) / F(np.sqrt(bc2)) + F(eps)).astype(F)
    return (p - F(lr / bc1) * (m / denom)).astype(F), m, v


def expon_lr(step, lr_init, lr_final, max_steps, delay_steps=0, delay_mult=1.0):
    """utils.py:7-44."""
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    rate = 1.0
    if delay_steps > 0:
        rate = delay_mult + (1 - delay_mult) * np.sin(0.5 * np.pi * np.clip(step / delay_steps, 0, 1))
    t = np.clip(step / max_steps, 0, 1)
    return rate * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)
