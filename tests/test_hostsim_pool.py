"""Pooling kernel SOURCES (csrc/pool.hip) on the hostsim emulator vs torch (adaptive max/avg mix, MaxPool2d(3,2,1))."""
import pytest

import kernel_cases as kc


@pytest.fixture(scope="module")
def hostsim_lib():
    return kc.build_hostsim()


@pytest.mark.parametrize("B,H,W,C,ties", [(2, 14, 14, 64, False), (3, 5, 7, 132, False), (2, 6, 6, 8, True), (1, 1, 1, 4, False)])
def test_mmpool(hostsim_lib, B, H, W, C, ties):
    kc.mmpool_case(hostsim_lib, "cpu", B, H, W, C, p0=0.3 if C != 8 else -1.2, seed=C, ties=ties)


@pytest.mark.parametrize("N,H,W,C,ties", [(2, 12, 12, 16, False), (1, 9, 7, 8, True), (2, 5, 6, 4, True), (1, 1, 1, 4, False)])
def test_maxpool3x3s2(hostsim_lib, N, H, W, C, ties):
    kc.maxpool_case(hostsim_lib, "cpu", N, H, W, C, seed=H, ties=ties)


@pytest.mark.parametrize("N,H,W,C,ksp", [(2, 12, 12, 48, (3, 1, 1)), (1, 5, 5, 64, (2, 2, 0)), (2, 7, 6, 8, (3, 1, 1)), (1, 4, 6, 4, (2, 2, 0))])
def test_maxpool_light_estimator_shapes(hostsim_lib, N, H, W, C, ksp):
    kc.maxpool_case(hostsim_lib, "cpu", N, H, W, C, seed=H + C, ties=True, ksp=ksp)
