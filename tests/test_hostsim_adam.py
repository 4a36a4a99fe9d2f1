"""Fused Adam kernel source on the hostsim emulator vs torch.optim.Adam (plain PyTorch fp32 reference)."""
import numpy as np
import pytest
import torch

import kernel_cases as kc


@pytest.fixture(scope="module")
def hostsim_lib():
    return kc.build_hostsim()


@pytest.mark.parametrize("n,wd", [(4096, 0.0), (1003, 0.01)])
def test_adam_matches_torch(hostsim_lib, n, wd):
    kc.adam_case(hostsim_lib, "cpu", n=n, wd=wd, steps=4)
