#!/usr/bin/env python3
"""Generate fixture G8 (densification + optimizer surgery) by
IMPORTING the reference's ``gsplat/gsmodel.py`` in the build container and
running it with CPU torch.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden_density.py

The reference hard-codes ``device="cuda"`` in one ``torch.zeros`` call
(gsmodel.py:272) and draws the split offsets with the device RNG
(``torch.normal``, gsmodel.py:274).  The harness below redirects that one
allocation to the CPU and replaces ``torch.normal`` by ``mean + std * noise``
with a seeded unit-normal ``noise`` that is stored in the fixture, so the
reference's arithmetic is untouched and reproducible.  Only data is written.

Reference functions exercised: ``GSModel.update_density_info`` (gsmodel.py:214-230),
``GSModel.update_gaussian_density`` (232-317), ``prune_params`` (151-166),
``update_params`` (132-148), ``GSModel.reset_alpha`` (319-330), with a real
``torch.optim.Adam`` (train.py:32) that has taken two steps.
"""
import os
import sys
import types

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
ply = types.ModuleType("plyfile"); ply.PlyData = object
sys.modules["plyfile"] = ply
sys.modules["gsplatcu"] = types.ModuleType("gsplatcu")
sys.path.insert(0, REF)
sys.path.insert(1, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import gsplat.gsmodel as ref_m  # noqa: E402
from tests.golden import _recipe  # noqa: E402

_recipe.assert_reference(ref_m)             # NOT compat/gsplat/gsmodel.py (the code under test)

NAMES = ("pws", "low_shs", "high_shs", "alphas_raw", "scales_raw", "rots_raw")
LRS = (0.001, 0.001, 0.001 / 20, 0.05, 0.005, 0.001)      # gsmodel.py:114-127


def make_params(n, rng):
    p = {
        "pws": rng.normal(0, 1, (n, 3)),
        "low_shs": rng.normal(0, 0.5, (n, 3)),
        "high_shs": rng.normal(0, 0.1, (n, 45)),
        # logit(0.005) = -5.29: about 15 % fall below and are pruned
        "alphas_raw": rng.normal(-2.0, 3.2, (n, 1)),
        # log(0.01) = -4.6 (clone/split boundary), log(0.1) = -2.3 (prune boundary)
        "scales_raw": rng.uniform(-7.0, -2.0, (n, 1)) + rng.normal(0, 0.25, (n, 3)),
        "rots_raw": rng.normal(0, 1, (n, 4)) * rng.uniform(0.3, 3.0, (n, 1)),
    }
    return {k: torch.tensor(v, dtype=torch.float32).requires_grad_() for k, v in p.items()}


def main():
    rng = np.random.default_rng(20240808)
    n, views = 400, 3
    params = make_params(n, rng)
    out = {"in_" + k: v.detach().numpy().copy() for k, v in params.items()}
    opt = torch.optim.Adam([{"params": [params[k]], "lr": lr, "name": k} for k, lr in zip(NAMES, LRS)],
                           lr=0.0, eps=1e-15)
    # two optimizer steps so that exp_avg / exp_avg_sq / step exist
    for s in range(2):
        for k in NAMES:
            g = torch.tensor(rng.normal(0, 1e-3, tuple(params[k].shape)), dtype=torch.float32)
            out["adam_grad%d_%s" % (s, k)] = g.numpy().copy()
            params[k].grad = g
        opt.step()
        opt.zero_grad(set_to_none=True)
    for k, grp in zip(NAMES, opt.param_groups):
        st = opt.state[grp["params"][0]]
        out["pre_" + k] = params[k].detach().numpy().copy()
        out["pre_m_" + k] = st["exp_avg"].numpy().copy()
        out["pre_v_" + k] = st["exp_avg_sq"].numpy().copy()
        out["pre_step_" + k] = np.array(float(st["step"]))

    model = ref_m.GSModel(1.0, 1000)
    # density statistics over three views (gsmodel.py:214-230)
    for v in range(views):
        mask = torch.tensor(rng.uniform(size=n) < 0.7)
        dus = torch.tensor(rng.normal(0, 1, (n, 2)) * np.exp(rng.uniform(-17.5, -13.0, (n, 1))), dtype=torch.float32)
        dus[~mask] = 0          # culled Gaussians receive no gradient
        out["view%d_dus" % v] = dus.numpy().copy()
        out["view%d_mask" % v] = mask.numpy().copy()
        model.us = types.SimpleNamespace(grad=dus)
        model.mask = mask
        model.update_density_info()
    out["grad_accum"] = model.grad_accum.numpy().copy()
    out["cunt"] = model.cunt.numpy().copy()

    # deterministic stand-ins for the two device-specific calls
    noise_full = rng.standard_normal((n, 3)).astype(np.float32)     # indexed by ORIGINAL Gaussian index
    out["unit_noise"] = noise_full
    real_zeros, real_normal = torch.zeros, torch.normal
    state = {}

    def zeros(*a, **k):
        k.pop("device", None)
        return real_zeros(*a, **k)

    def normal(mean, std):
        return mean + std * state["noise"]

    # the split selection (needed to index the noise the way the reference orders its samples) is
    # recomputed here from the reference's own helpers
    with torch.no_grad():
        sel_small = params["alphas_raw"].squeeze() < ref_m.get_alphas_raw(model.alpha_threshold)
        sel_big = torch.max(params["scales_raw"], axis=1)[0] > ref_m.get_scales_raw(model.big_threshold)
        remain = ~(sel_small | sel_big)
        grads = model.grad_accum.squeeze()[remain] / model.cunt[remain]
        grads[grads.isnan()] = 0.0
        scales = ref_m.get_scales(params["scales_raw"][remain])
        by_grad = grads >= model.grad_threshold
        by_scale = torch.max(scales, axis=1)[0] <= model.scale_threshold
        split = by_grad & ~by_scale
        orig_idx = torch.arange(n)[remain][split]
        state["noise"] = torch.tensor(noise_full)[orig_idx]
        out["expect_remain"] = remain.numpy().copy()
        out["expect_clone"] = (torch.arange(n)[remain][by_grad & by_scale]).numpy().copy()
        out["expect_split"] = orig_idx.numpy().copy()

    torch.zeros, torch.normal = zeros, normal
    try:
        with torch.no_grad():
            model.update_gaussian_density(params, opt)
    finally:
        torch.zeros, torch.normal = real_zeros, real_normal
    for k, grp in zip(NAMES, opt.param_groups):
        st = opt.state[grp["params"][0]]
        out["post_" + k] = params[k].detach().numpy().copy()
        out["post_m_" + k] = st["exp_avg"].numpy().copy()
        out["post_v_" + k] = st["exp_avg_sq"].numpy().copy()
        out["post_step_" + k] = np.array(float(st["step"]))
    assert model.grad_accum is None and model.cunt is None

    with torch.no_grad():
        model.reset_alpha(params, opt)
    grp = opt.param_groups[3]
    st = opt.state[grp["params"][0]]
    out["reset_alphas_raw"] = params["alphas_raw"].detach().numpy().copy()
    out["reset_m_alphas_raw"] = st["exp_avg"].numpy().copy()
    out["reset_v_alphas_raw"] = st["exp_avg_sq"].numpy().copy()

    # one more Adam step on the densified model (new rows start from zero moments, old step count)
    for k in NAMES:
        g = torch.tensor(rng.normal(0, 1e-3, tuple(params[k].shape)), dtype=torch.float32)
        out["adam_grad2_%s" % k] = g.numpy().copy()
        params[k].grad = g
    opt.step()
    for k, grp in zip(NAMES, opt.param_groups):
        st = opt.state[grp["params"][0]]
        out["final_" + k] = params[k].detach().numpy().copy()
        out["final_m_" + k] = st["exp_avg"].numpy().copy()
        out["final_v_" + k] = st["exp_avg_sq"].numpy().copy()

    # learning-rate schedule samples (gsmodel.py:180-183, 332-338; utils.py:7-44)
    sched = ref_m.get_expon_lr_func(lr_init=1e-4 * 2.5, lr_final=1e-6 * 2.5, lr_delay_mult=0.01, max_steps=3000)
    steps = np.array([0, 1, 10, 1500, 2999, 3000, 4000])
    out["lr_steps"] = steps
    out["lr_values"] = np.array([sched(int(s)) for s in steps])

    doc = ("G8: gsplat/gsmodel.py densification on %d Gaussians, CPU torch %s; thresholds of GSModel(1.0, .): "
           "alpha 0.005, big 0.1, scale 0.01, grad 4e-7.  torch.normal replaced by mean + std*unit_noise[orig_idx]."
           % (n, torch.__version__))
    _recipe.save("g8_densify.npz", doc, **out)
    print("remain %d clone %d split %d -> %d" % (int(out["expect_remain"].sum()), len(out["expect_clone"]),
                                                 len(out["expect_split"]), out["post_pws"].shape[0]))


if __name__ == "__main__":
    _recipe.begin("--check" in sys.argv[1:])
    main()
    sys.exit(_recipe.finish())
