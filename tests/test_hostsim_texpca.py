"""Texture-PCA decode kernel SOURCES (csrc/texpca.hip) on the hostsim emulator vs torch."""
import pytest

import kernel_cases as kc


@pytest.fixture(scope="module")
def hostsim_lib():
    return kc.build_hostsim()


@pytest.mark.parametrize("B,K,n,with_mean", [(3, 10, 2336, True), (17, 10, 4096, False), (1, 32, 1028, True), (20, 7, 8, True)])
def test_texture_pca(hostsim_lib, B, K, n, with_mean):
    kc.texture_pca_case(hostsim_lib, "cpu", B, K, n, seed=B + n, with_mean=with_mean)
