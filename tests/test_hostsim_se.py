"""Squeeze-and-excitation kernel SOURCES (csrc/se.hip, csrc/mlp.hip act 2 / 3) on the hostsim emulator vs torch autograd."""
import pytest

import kernel_cases as kc


@pytest.fixture(scope="module")
def hostsim_lib():
    return kc.build_hostsim()


@pytest.mark.parametrize("B,H,W,C,SQ", [(2, 6, 6, 40, 10), (3, 5, 4, 144, 6), (2, 9, 9, 288, 12), (1, 3, 3, 816, 34), (35, 2, 2, 72, 5)])
def test_squeeze_excite(hostsim_lib, B, H, W, C, SQ):
    kc.se_case(hostsim_lib, "cpu", B, H, W, C, SQ, seed=C)


def test_drop_connect_add(hostsim_lib):
    """drop-connect + skip of an MBConv block in one pass (csrc/se.hip) vs the reference's expression (utils.py:82-91)."""
    import torch
    gen = torch.Generator().manual_seed(3)
    B, n, keep = 5, 7 * 6 * 24, 0.8
    x, skip, u = torch.randn(B, n, generator=gen), torch.randn(B, n, generator=gen), torch.rand(B, generator=gen)
    out = torch.empty(B, n)
    hostsim_lib.drop_connect_add(x, skip, u, keep, B, n, out)
    ref = x / keep * torch.floor(keep + u)[:, None] + skip
    assert float((out - ref).abs().max()) <= 1e-6 and float(torch.floor(keep + u).min()) == 0.0 and float(torch.floor(keep + u).max()) == 1.0
    hostsim_lib.drop_connect_add(x, None, u, keep, B, n, out)
    assert float((out - x / keep * torch.floor(keep + u)[:, None]).abs().max()) <= 1e-6
