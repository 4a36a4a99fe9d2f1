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


@pytest.mark.parametrize("grid", ["1", "3"])
def test_ssim_persistent_workgroups_walk_several_tiles(hostsim_lib, golden_dir, monkeypatch, grid):
    """The kernels are persistent (round 5): with fewer workgroups than tiles each one walks several, the next tile's halo held in registers
    meanwhile -- same results (partial sums are written per tile, in tile order)."""
    monkeypatch.setenv("HIFIHR_SSIM_GRID", grid)
    g = np.load(os.path.join(golden_dir, "ssim.npz"))
    kc.ssim_case(hostsim_lib, "cpu", g["a"], g["b"], g["ssim"], g["ga"])
    gen = torch.Generator().manual_seed(9)
    a = torch.rand(2, 3, 70, 45, generator=gen)
    b = (a + 0.3 * torch.rand(2, 3, 70, 45, generator=gen)).clamp(0, 1)
    kc.ssim_case(hostsim_lib, "cpu", a.numpy(), b.numpy())
