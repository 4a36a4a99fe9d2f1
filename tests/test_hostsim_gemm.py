"""Batched f32 GEMM kernel SOURCES (csrc/gemm.hip: 16x16x4 MFMA, direct-to-LDS operand images) on the hostsim emulator vs torch."""
import pytest

import kernel_cases as kc


@pytest.fixture(scope="module")
def hostsim_lib():
    return kc.build_hostsim()


@pytest.mark.parametrize("M,N,K,batch", [(128, 128, 32, 1), (98, 128, 64, 2), (200, 64, 96, 1), (130, 192, 32, 2), (40, 256, 64, 1)])
def test_bgemm_nt(hostsim_lib, M, N, K, batch):
    kc.bgemm_case(hostsim_lib, "cpu", M, N, K, batch, seed=M + N)


@pytest.mark.parametrize("M,N,T,batch", [(128, 128, 64, 1), (64, 128, 98, 2), (128, 64, 40, 1), (64, 64, 33, 2), (256, 128, 320, 1), (64, 64, 777, 1)])
def test_bgemm_tn(hostsim_lib, M, N, T, batch):
    kc.bgemm_tn_case(hostsim_lib, "cpu", M, N, T, batch, seed=M + T)
