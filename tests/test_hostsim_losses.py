"""Fused loss kernel SOURCES (csrc/losses.hip) on the hostsim emulator vs the torch-op restatement of the reference's
LossFunction terms (hifihr_amd/losses.py, itself pinned by tests/golden/losses.npz)."""
import pytest

import kernel_cases as kc


@pytest.fixture(scope="module")
def hostsim_lib():
    return kc.build_hostsim()


@pytest.mark.parametrize("B,V,F,mse", [(3, 50, 80, False), (2, 778, 1538, True), (1, 12, 0, False), (4, 30, 41, True)])
def test_geom_losses(hostsim_lib, B, V, F, mse):
    kc.geom_loss_case(hostsim_lib, "cpu", B, V, F, mse, seed=V + F)


@pytest.mark.parametrize("B,H,W,with_g", [(2, 16, 16, True), (3, 12, 20, False), (1, 4, 4, True)])
def test_photo_losses(hostsim_lib, B, H, W, with_g):
    kc.photo_loss_case(hostsim_lib, "cpu", B, H, W, seed=H * W, with_g=with_g)


@pytest.mark.parametrize("B,mse,use2,use3", [(3, False, True, True), (48, True, True, False), (2, False, False, True)])
def test_joint_terms(hostsim_lib, B, mse, use2, use3):
    kc.joint_terms_case(hostsim_lib, "cpu", B, mse, seed=B, use2=use2, use3=use3)


def test_loss_total_kernels(hostsim_lib):
    kc.loss_total_case(hostsim_lib, "cpu")


def test_light_split_kernels(hostsim_lib):
    kc.light_split_case(hostsim_lib, "cpu")
    kc.light_split_case(hostsim_lib, "cpu", B=300, seed=3)
