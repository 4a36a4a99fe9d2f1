"""Runs the MANO HIP kernel SOURCES on the CPU through tests/hostsim (a HIP execution-model emulator) and
checks them against the reference-pinned golden vectors and the oracle.  The real parity tests are the
`gpu`-marked ones in test_gpu_mano.py; this module exists so kernel bugs are found without a GPU."""
import os

import numpy as np
import pytest

import kernel_cases as kc


@pytest.fixture(scope="module")
def hostsim_lib():
    return kc.build_hostsim()


def test_mano_kernels_vs_reference_golden(hostsim_lib, synth_tables, golden_dir):
    g = dict(np.load(os.path.join(golden_dir, "mano_synth.npz")))
    kc.mano_fwd_bwd_case(hostsim_lib, synth_tables, g, "cpu")


def test_mano_kernels_vs_oracle_random(hostsim_lib, synth_tables):
    kc.mano_random_vs_oracle_case(hostsim_lib, synth_tables, "cpu", B=3, seed=5)


@pytest.mark.parametrize("root_id", [9, 0, -1])
def test_mano_joints_kernels(hostsim_lib, synth_tables, root_id):
    kc.mano_joints_case(hostsim_lib, synth_tables, "cpu", B=2, seed=3, root_id=root_id)


@pytest.mark.parametrize("B,root_id,with_cam", [(3, 9, True), (2, 0, False), (2, -1, True)])
def test_mano_full_kernels(hostsim_lib, synth_tables, B, root_id, with_cam):
    kc.mano_full_case(hostsim_lib, synth_tables, "cpu", B=B, seed=41 + B, root_id=root_id, with_cam=with_cam)
