"""Pins oracle/mano_oracle.py against vectors produced by the reference's own ManoLayer
(tools/make_golden.py; reference utils/my_mano.py:315-483, utils/manopth/rodrigues_layer.py:43-54)."""
import os

import numpy as np
import pytest
import torch

from conftest import mano_pkl_path
from oracle import mano_oracle as mo


def _check(tables, g):
    pose = torch.tensor(g["pose"], requires_grad=True)
    beta = torch.tensor(g["beta"], requires_grad=True)
    verts, jtr, _ = mo.mano_forward(tables, pose, beta)
    ((verts * torch.tensor(g["wv"])).sum() + (jtr * torch.tensor(g["wj"])).sum()).backward()
    np.testing.assert_allclose(verts.detach().numpy(), g["verts"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(jtr.detach().numpy(), g["jtr"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(pose.grad.numpy(), g["gpose"], atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(beta.grad.numpy(), g["gbeta"], atol=2e-4, rtol=1e-4)


def test_mano_oracle_vs_reference_synthetic_tables(golden_dir, synth_tables):
    _check(synth_tables, np.load(os.path.join(golden_dir, "mano_synth.npz")))


def test_mano_oracle_vs_reference_real_tables(golden_dir):
    pkl = mano_pkl_path()
    if pkl is None:
        pytest.skip("MANO_RIGHT.pkl is not redistributable; set HIFIHR_MANO_PKL to run")
    path = os.path.join(golden_dir, "mano_real.npz")
    if not os.path.exists(path):
        # posed real-MANO meshes are data derived from the licensed model: NOT committed ("MANO licence").
        # A licence holder regenerates them next to their own MANO_RIGHT.pkl with `python tools/make_golden.py`.
        pytest.skip("tests/golden/mano_real.npz is not shipped (derived MANO data); regenerate it with tools/make_golden.py")
    from hifihr_amd.mano_tables import load_mano_pkl
    _check(load_mano_pkl(pkl), np.load(path))


def test_rodrigues_oracle_vs_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "rodrigues.npz"))
    aa = torch.tensor(g["aa"], requires_grad=True)
    rot = mo.batch_rodrigues(aa)
    (rot * torch.tensor(g["w"])).sum().backward()
    np.testing.assert_allclose(rot.detach().numpy(), g["rot"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(aa.grad.numpy(), g["gaa"], atol=1e-4, rtol=1e-4)


def test_center_joint_is_zero(synth_tables):
    pose = 0.3 * torch.randn(3, 48)
    beta = 0.3 * torch.randn(3, 10)
    _, jtr, _ = mo.mano_forward(synth_tables, pose, beta)
    assert float(jtr[:, 9].abs().max()) == 0.0


def test_xyz_from_vertice_layout(synth_tables):
    verts = torch.randn(2, 778, 3)
    j = mo.xyz_from_vertice(synth_tables, verts)
    assert j.shape == (2, 21, 3)
    # tips are copied vertices (Freihand_trainer_mano_fullsup.py:177-183)
    for slot, vid in mo.XYZ_TIPS.items():
        assert torch.equal(j[:, slot], verts[:, vid])
    jr = torch.tensor(synth_tables.J_regressor)
    np.testing.assert_allclose(j[:, 9].numpy(), (jr[4] @ verts).numpy(), atol=1e-5)   # manoId 4 -> slot 9


def test_synthetic_tables_are_mano_shaped(synth_tables):
    t = synth_tables
    t.check()
    assert np.allclose(t.weights.sum(1), 1, atol=1e-5) and (np.count_nonzero(t.weights, axis=1) <= 4).all()
    assert np.allclose(t.J_regressor.sum(1), 1, atol=1e-5)
    # disc topology with a 16-edge boundary: every edge in <=2 faces, exactly 16 in one
    e = np.sort(np.concatenate([t.faces[:, [0, 1]], t.faces[:, [1, 2]], t.faces[:, [2, 0]]]), axis=1)
    _, cnt = np.unique(e, axis=0, return_counts=True)
    assert cnt.max() == 2 and int((cnt == 1).sum()) == 16
