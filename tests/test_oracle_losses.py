"""oracle/loss_oracle.py (restatement of reference losses.py:226-453) against vectors produced by the reference's OWN
LossFunction.__call__ (tests/golden/loss_dict.npz, tools/make_golden.py:gen_loss_dict), and the model-tail restatements against
models_res_nimble.py:209-220 / :228-235 executed from source (tests/golden/model_tail.npz).  CPU only."""
import os

import numpy as np
import pytest
import torch

import loss_cases
from oracle import loss_oracle as lo
from oracle import render_oracle as ro
from oracle.torch_modules import PerceptualLossRef


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "loss_dict.npz"))


@pytest.mark.parametrize("name", ["cfg2", "cfg3", "ho3d"])
def test_loss_oracle_matches_reference_loss_function(golden, name):
    args, ex, out, dat = loss_cases.loss_dict_case(name)
    leaves = {}
    if name != "cfg3":
        for k in ("re_img", "joints", "mano_verts", "pose_params", "shape_params"):
            out[k] = out[k].clone().requires_grad_(True)
            leaves[k] = out[k]
    d = lo.LossFunctionRef(PerceptualLossRef())(ex, out, args.losses, dat, args)
    assert sorted(d.keys()) == list(golden[f"{name}/keys"]), (sorted(d.keys()), list(golden[f"{name}/keys"]))
    for k, v in d.items():
        ref = float(golden[f"{name}/{k}"])
        assert abs(float(v) - ref) <= 1e-6 * abs(ref) + 1e-12, (name, k, float(v), ref)
    if leaves:
        sum(d[k] for k in args.losses if k in d).backward()
        for k, t in leaves.items():
            key = f"{name}/grad/{k}"
            if key in golden.files:
                g = golden[key]
                np.testing.assert_allclose(t.grad.numpy(), g, rtol=1e-5, atol=1e-7 * np.abs(g).max())


def test_model_tail_restatements_match_reference_lines(golden_dir):
    g = np.load(os.path.join(golden_dir, "model_tail.npz"))
    rgba = torch.from_numpy(g["rgba"]); images = torch.from_numpy(g["images"])
    # the oracle's resolve takes [B,4,R,R] sub-sample planes; the reference permutes NHWC -> NCHW first (:210)
    resolved = torch.nn.functional.avg_pool2d(rgba.permute(0, 3, 1, 2), 3, 3)
    re_img, re_sil, mask_rgbs = ro.model_render_outputs(resolved, images)
    np.testing.assert_array_equal(re_sil.numpy(), g["re_sil"])
    np.testing.assert_allclose(re_img.numpy(), g["re_img"], rtol=0, atol=1e-7)
    np.testing.assert_array_equal(mask_rgbs.numpy(), g["maskRGBs"])
    cam = ro.ndc_camera_from_K(torch.from_numpy(g["Ks"]), 224.0)      # [B,4] = (fx, fy, px, py) in NDC
    np.testing.assert_allclose(-cam[:, :2].numpy(), g["focal"], rtol=1e-7)      # the oracle folds in `focal_length=-fcl` (:184-186)
    np.testing.assert_allclose(cam[:, 2:].numpy(), g["principal"], rtol=1e-7, atol=1e-7)
