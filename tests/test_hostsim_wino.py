"""Winograd F(2x2, 3x3) kernel SOURCES (csrc/wino.hip + batched conv_igemm) on the hostsim emulator vs torch conv2d."""
import pytest

import kernel_cases as kc


@pytest.fixture(scope="module")
def hostsim_lib():
    return kc.build_hostsim()


@pytest.mark.parametrize("N,H,W,C,K", [(2, 6, 6, 32, 64), (1, 7, 5, 64, 32), (3, 4, 4, 32, 96), (1, 5, 6, 128, 64), (2, 4, 4, 128, 128)])
def test_winograd_fwd_bwd(hostsim_lib, N, H, W, C, K):
    kc.wino_case(hostsim_lib, "cpu", N, H, W, C, K, seed=C + K)


def test_weight_prep_equals_separate_transforms(hostsim_lib):
    kc.weight_prep_case(hostsim_lib, "cpu")
