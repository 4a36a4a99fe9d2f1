"""CPU: oracle/torch_modules.{HandEncoderRef, LightEstimatorRef, MMPoolRef} -- what the GPU tests, the whole-step tests and OracleModel
compare the HIP heads with -- against tests/golden/heads.npz, i.e. against the REFERENCE's own classes (network/res_encoder.py:53-167,
:169-209, :247-265) run from source by tools/make_golden.gen_heads on the same recipe (tests/heads_fixture.py).  The restatements
are the same torch ops in the same order: the expected difference is 0; the bound only leaves room for another host's BLAS blocking."""
import os

import numpy as np
import pytest
import torch

import heads_fixture as hf

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "heads.npz")


def _ref_module(name):
    from oracle import torch_modules as tm
    cls, cargs, _ = hf.CASES[name]
    return {"HandEncoder": tm.HandEncoderRef, "LightEstimator": tm.LightEstimatorRef, "MMPool": tm.MMPoolRef}[cls](*cargs)


# In training mode the bias of a Linear that feeds a BatchNorm1d has NO gradient (the batch mean is subtracted again): what either side
# computes for it is rounding noise around zero (~1e-6 here, changing with the order of the float atomics from run to run).  These entries
# are compared against an absolute floor instead of a bound relative to their own (noise) magnitude.
NOISE_GRADS_TRAIN = ("base_layers.0.bias", "base_layers.3.bias")
NOISE_FLOOR = 2e-5


def compare_with_fixture(module, name, train, device, rtol_out, rtol_grad, gold=None, skip_grads=()):
    """Every stored array of one (case, mode) against a module of the same state-dict layout; returns the worst ratio observed / bound."""
    g = gold if gold is not None else np.load(GOLD)
    tag = f"{name}/{'train' if train else 'eval'}"
    keys = [k for k in g.files if k.startswith(tag + "/")]
    assert keys, tag
    hf.fill_state(module, name)
    outs, grads, bufs = hf.run_case(module, name, train, device)
    worst = 0.0
    worst_key = [""]
    seen = set()
    for k in keys:
        kind, sub = k[len(tag) + 1:].split("/", 1)
        want = g[k]
        if kind == "out":
            got, tol = outs[sub].detach().cpu().numpy(), rtol_out
        elif kind == "grad":
            if sub in skip_grads:
                continue
            assert sub in grads, f"{tag}: the reference has a gradient for {sub}, the module does not"
            got, tol = hf.sample(grads[sub]), rtol_grad
        else:
            got, tol = bufs[sub].detach().cpu().numpy(), rtol_out
        seen.add((kind, sub))
        assert got.shape == want.shape, (k, got.shape, want.shape)
        bound = tol * max(float(np.abs(want).max()), 1e-3)
        if kind == "grad" and train and name.startswith("he_") and sub in NOISE_GRADS_TRAIN:
            assert float(np.abs(want).max()) <= NOISE_FLOOR, (k, float(np.abs(want).max()))      # the reference's own value is noise too
            bound = NOISE_FLOOR
        err = float(np.abs(got - want).max())
        assert err <= bound, f"{k}: |diff| {err:.3e} > {bound:.3e}"
        if err / bound > worst:
            worst, worst_key[0] = err / bound, k
    # nothing the module produces may be missing from the reference's record either (e.g. an extra head)
    for sub in outs:
        assert ("out", sub) in seen, f"{tag}: output {sub} is not in the reference's record"
    print(f"[margin] {tag}: worst observed / bound {worst:.3g} at {worst_key[0]}")
    return worst


@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("name", list(hf.CASES))
def test_oracle_heads_match_reference_fixture(name, train):
    torch.manual_seed(0)
    worst = compare_with_fixture(_ref_module(name), name, train, "cpu", 1e-6, 1e-6)
    print(f"{name} train={train}: worst observed / bound = {worst:.3g}")


def test_fixture_holds_every_case_and_mode():
    g = np.load(GOLD)
    for name in hf.CASES:
        for mode in ("train", "eval"):
            assert any(k.startswith(f"{name}/{mode}/out/") for k in g.files)
            assert f"{name}/{mode}/grad/x" in g.files
    # the mano encoder has no texture head and the nimble one no rotation head (res_encoder.py:98-104, :118-125)
    assert not any("he_mano512/train/out/texture_params" in k for k in g.files)
    assert not any("he_nimble1536/train/out/rot" in k for k in g.files)
