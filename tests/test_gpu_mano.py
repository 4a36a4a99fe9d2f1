"""GPU parity: MANO HIP kernels through the C ABI vs reference-produced golden vectors and the oracle."""
import os

import numpy as np
import pytest
import torch

import kernel_cases as kc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from hifihr_amd._lib import get_lib
    assert torch.cuda.is_available()
    return get_lib()


def test_mano_vs_reference_golden(lib, synth_tables, golden_dir):
    g = dict(np.load(os.path.join(golden_dir, "mano_synth.npz")))
    kc.mano_fwd_bwd_case(lib, synth_tables, g, "cuda")


@pytest.mark.parametrize("B", [1, 32, 256])
def test_mano_vs_oracle_random(lib, synth_tables, B):
    kc.mano_random_vs_oracle_case(lib, synth_tables, "cuda", B=B, seed=100 + B)


@pytest.mark.parametrize("root_id", [9, 0, -1])
def test_mano_joints(lib, synth_tables, root_id):
    kc.mano_joints_case(lib, synth_tables, "cuda", B=33, seed=3, root_id=root_id)


@pytest.mark.parametrize("B,root_id,with_cam", [(1, 9, True), (32, 9, True), (33, 0, True), (256, 9, False), (5, -1, True)])
def test_mano_full_one_launch_per_direction(lib, synth_tables, B, root_id, with_cam):
    """The step's form of the MANO rows (A7 + A8 + A10): layer + joints + root-relative + camera offset in one launch each way."""
    kc.mano_full_case(lib, synth_tables, "cuda", B=B, seed=700 + B, root_id=root_id, with_cam=with_cam)


def test_mano_bwd_is_deterministic(lib, synth_tables):
    h = lib.mano_create(synth_tables)
    B = 64
    pose = 0.5 * torch.randn(B, 48, device="cuda")
    beta = 0.5 * torch.randn(B, 10, device="cuda")
    verts = torch.empty(B, 778, 3, device="cuda"); jtr = torch.empty(B, 21, 3, device="cuda"); saved = torch.empty_like(verts)
    lib.mano_lbs_fwd(h, pose, beta, verts, jtr, saved)
    gv = torch.randn_like(verts); gj = torch.randn_like(jtr)
    outs = []
    for _ in range(2):
        gp = torch.empty(B, 48, device="cuda"); gb = torch.empty(B, 10, device="cuda")
        lib.mano_lbs_bwd(h, pose, beta, saved, gv, gj, gp, gb)
        outs.append((gp.clone(), gb.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    lib.mano_destroy(h)
