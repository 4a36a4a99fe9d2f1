"""Squeeze-and-excitation kernel SOURCES (csrc/se.hip, csrc/mlp.hip act 2 / 3) on the hostsim emulator vs torch autograd."""
import pytest

import kernel_cases as kc


@pytest.fixture(scope="module")
def hostsim_lib():
    return kc.build_hostsim()


@pytest.mark.parametrize("B,H,W,C,SQ", [(2, 6, 6, 40, 10), (3, 5, 4, 144, 6), (2, 9, 9, 288, 12), (1, 3, 3, 816, 34), (35, 2, 2, 72, 5)])
def test_squeeze_excite(hostsim_lib, B, H, W, C, SQ):
    kc.se_case(hostsim_lib, "cpu", B, H, W, C, SQ, seed=C)
