"""csrc/eval.hip on the host emulator vs the reference's align_w_scale golden vectors (tests/golden/eval.npz)."""
import pytest

import kernel_cases as kc


@pytest.fixture(scope="module")
def hostsim_lib():
    return kc.build_hostsim()


def test_procrustes_alignment_vs_reference(hostsim_lib, golden_dir):
    kc.procrustes_case(hostsim_lib, "cpu", golden_dir)


def test_ho3d_joint_maps_vs_reference(golden_dir):
    import os
    import numpy as np
    import torch
    from hifihr_amd.traineval import Frei2HO3D, HO3D2Frei
    g = np.load(os.path.join(golden_dir, "eval.npz"))
    j = torch.from_numpy(g["j"])
    assert np.array_equal(HO3D2Frei(j).numpy(), g["ho3d2frei"]) and np.array_equal(Frei2HO3D(j).numpy(), g["frei2ho3d"])
