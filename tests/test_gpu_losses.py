"""The product LossFunction (fused HIP kernels) on the GPU against vectors produced by the reference's OWN LossFunction.__call__
(losses.py:226-453 executed from source by tools/make_golden.py -> tests/golden/loss_dict.npz): every term of the loss lists of
BASELINE configs[1] (L2 base loss), configs[2] (+ scale / mscale / iou) and configs[4] (HO-3D list + the self-supervised terms),
and the gradient of the summed loss w.r.t. the render, joints, vertices, pose and shape.  Tolerance: 1e-4 (north_star)."""
import os

import numpy as np
import pytest
import torch

import loss_cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["cfg2", "cfg3", "ho3d"])
def test_loss_function_matches_reference_vectors(golden_dir, name):
    from hifihr_amd.losses import LossFunction
    g = np.load(os.path.join(golden_dir, "loss_dict.npz"))
    args, ex, out, dat = loss_cases.loss_dict_case(name)
    dev = torch.device("cuda")
    ex = {k: v.to(dev) for k, v in ex.items()}
    ex["segms_gt"] = ex["segms_gt"].long()                    # data_dic hands the mask over as int64 (utils/traineval_util.py:58)
    out = {k: v.to(dev) for k, v in out.items()}
    leaves = {}
    if name != "cfg3":
        for k in ("re_img", "joints", "mano_verts", "pose_params", "shape_params"):
            out[k] = out[k].clone().requires_grad_(True)
            leaves[k] = out[k]
    d = LossFunction()(ex, out, args.losses, dat, args)
    assert sorted(d.keys()) == list(g[f"{name}/keys"]), (sorted(d.keys()), list(g[f"{name}/keys"]))
    report = {k: (float(v.detach()), float(g[f"{name}/{k}"])) for k, v in d.items()}
    print(name, report)
    for k, (a, b) in report.items():
        assert abs(a - b) <= 1e-4 * abs(b) + 1e-9, (name, k, a, b)
    if leaves:
        sum(d[k] for k in args.losses if k in d).backward()
        torch.cuda.synchronize()
        for k, t in leaves.items():
            key = f"{name}/grad/{k}"
            if key in g.files:
                ref = g[key]
                err = np.abs(t.grad.cpu().numpy() - ref).max()
                assert err <= 2e-4 * np.abs(ref).max() + 1e-12, (name, k, err, np.abs(ref).max())
        # the step's form of the sum: LossFunction.total (one launch each way when the requested terms are exactly the fused kernels'
        # vectors -- cfg2 -- else stack + sum): the same value and the same gradients
        for t in leaves.values():
            t.grad = None
        lf = LossFunction()
        d2 = lf(ex, out, args.losses, dat, args)
        if all(k in d2 for k in args.losses):
            tot = lf.total(d2, args.losses)
            want = sum(float(d2[k].detach()) for k in args.losses)
            assert abs(float(tot.detach()) - want) <= 1e-6 * max(1.0, abs(want)), (float(tot.detach()), want)
            tot.backward()
            torch.cuda.synchronize()
            for k, t in leaves.items():
                key = f"{name}/grad/{k}"
                if key in g.files:
                    ref = g[key]
                    err = np.abs(t.grad.cpu().numpy() - ref).max()
                    assert err <= 2e-4 * np.abs(ref).max() + 1e-12, ("total", name, k, err, np.abs(ref).max())
