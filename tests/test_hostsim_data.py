"""csrc/augment.hip on the host emulator: bit-exact with the reference's PIL warp (tests/golden/data_path.npz)."""
import pytest

import kernel_cases as kc


@pytest.fixture(scope="module")
def hostsim_lib():
    return kc.build_hostsim()


def test_freihand_augment_vs_reference_pil(hostsim_lib, golden_dir):
    kc.augment_case(hostsim_lib, "cpu", golden_dir)


def test_freihand_batch_two_launches(hostsim_lib):
    kc.freihand_batch_case(hostsim_lib, "cpu", seed=1)
    kc.freihand_batch_case(hostsim_lib, "cpu", seed=2, B=1, J=21, V=778)


def test_ho3d_crop_resize_vs_pillow(hostsim_lib, golden_dir):
    kc.ho3d_batch_case(hostsim_lib, "cpu", golden_dir)
