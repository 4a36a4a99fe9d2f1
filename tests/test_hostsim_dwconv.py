"""Depthwise convolution kernel SOURCES on the hostsim emulator vs torch (grouped conv2d with static same padding)."""
import pytest

import kernel_cases as kc


@pytest.fixture(scope="module")
def hostsim_lib():
    return kc.build_hostsim()


@pytest.mark.parametrize("N,H,W,C,K,stride", [(2, 9, 7, 24, 3, 1), (1, 12, 12, 144, 3, 2), (2, 8, 8, 48, 5, 1), (1, 11, 9, 288, 5, 2),
                                             (1, 4, 4, 1392, 5, 1)])
def test_dwconv_fwd_bwd(hostsim_lib, N, H, W, C, K, stride):
    kc.dwconv_case(hostsim_lib, "cpu", N, H, W, C, K, stride, seed=C)


@pytest.mark.parametrize("N,H,W,C,K,stride", [(2, 9, 7, 24, 3, 1), (1, 12, 12, 144, 3, 2), (2, 8, 8, 48, 5, 1), (1, 11, 9, 288, 5, 2),
                                             (2, 4, 4, 816, 5, 1)])
def test_dwconv_with_batchnorm_and_swish_on_load(hostsim_lib, N, H, W, C, K, stride):
    kc.dwconv_bnswish_case(hostsim_lib, "cpu", N, H, W, C, K, stride, seed=C + 1)
