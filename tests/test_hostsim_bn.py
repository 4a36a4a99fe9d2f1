"""Fused BatchNorm(+add+ReLU) kernel SOURCES and the conv epilogue statistics on the hostsim emulator vs torch."""
import pytest

import kernel_cases as kc


@pytest.fixture(scope="module")
def hostsim_lib():
    return kc.build_hostsim()


@pytest.mark.parametrize("C,relu,residual", [(64, True, False), (128, True, True), (256, False, False), (16, True, True), (40, "swish", False), (1392, "swish", False),
                                               (136, False, False), (2304, True, True)])
def test_bn_act_fwd_bwd(hostsim_lib, C, relu, residual):
    kc.bn_act_case(hostsim_lib, "cpu", 3, 5, 7, C, relu, residual, seed=C)


@pytest.mark.parametrize("N,H,W,C", [(2, 8, 10, 64), (1, 7, 9, 16), (3, 5, 2, 8), (1, 12, 12, 256)])
def test_bn_relu_maxpool_stem(hostsim_lib, N, H, W, C):
    kc.bn_relu_maxpool_case(hostsim_lib, "cpu", N, H, W, C, seed=H + C)


def test_conv_epilogue_bn_statistics(hostsim_lib):
    kc.conv_bnstats_case(hostsim_lib, "cpu", 2, 9, 7, 16, 64, 3, 1, 1)
    kc.conv_bnstats_case(hostsim_lib, "cpu", 1, 16, 16, 4, 64, 7, 2, 3)


@pytest.mark.parametrize("producer", ["stats", "conv3x3", "conv1x1", "halo", "wino4", "wino2", "dw"])
def test_bn_statistics_with_mean_much_larger_than_std(hostsim_lib, producer):
    """mean 50 / std 0.1 channels: the shifted-sum / fp64-slot statistics of every producer against float64 statistics."""
    kc.bn_large_mean_case(hostsim_lib, "cpu", producer)
