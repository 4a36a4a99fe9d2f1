"""Fused SSIM kernel SOURCES on the hostsim emulator vs the reference's pytorch_ssim vectors."""
import os

import numpy as np
import pytest
import torch

import kernel_cases as kc


@pytest.fixture(scope="module")
def hostsim_lib():
    return kc.build_hostsim()


def test_ssim_vs_reference_golden(hostsim_lib, golden_dir):
    g = np.load(os.path.join(golden_dir, "ssim.npz"))
    kc.ssim_case(hostsim_lib, "cpu", g["a"], g["b"], g["ssim"], g["ga"])


def test_ssim_ragged_size(hostsim_lib):
    gen = torch.Generator().manual_seed(5)
    a = torch.rand(1, 2, 37, 21, generator=gen)
    b = (a + 0.3 * torch.rand(1, 2, 37, 21, generator=gen)).clamp(0, 1)
    kc.ssim_case(hostsim_lib, "cpu", a.numpy(), b.numpy())
