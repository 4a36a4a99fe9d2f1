"""Fully connected head kernel SOURCES (csrc/mlp.hip) on the hostsim emulator vs nn.Linear / BatchNorm1d / ReLU autograd."""
import pytest

import kernel_cases as kc


@pytest.fixture(scope="module")
def hostsim_lib():
    return kc.build_hostsim()


@pytest.mark.parametrize("B,I,O,act,bn,need_dx", [(32, 128, 48, 0, False, True), (6, 512, 40, 1, True, True), (4, 32, 3, 0, False, True),
                                                  (5, 128, 1, 0, False, False), (64, 72, 33, 1, True, True), (70, 36, 20, 1, False, True),
                                                  (3, 256, 300, 1, False, True), (5, 1100, 20, 1, False, True), (4, 1038, 6, 1, True, True)])
def test_linear(hostsim_lib, B, I, O, act, bn, need_dx):
    kc.linear_case(hostsim_lib, "cpu", B, I, O, act, bn, seed=B + O, need_dx=need_dx)


def test_grouped_linear_layers_equal_single_launches(hostsim_lib):
    kc.linear_group_case(hostsim_lib, "cpu")
