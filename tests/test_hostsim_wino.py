"""Winograd F(2x2, 3x3) kernel SOURCES (csrc/wino.hip + batched conv_igemm) on the hostsim emulator vs torch conv2d."""
import pytest

import kernel_cases as kc


@pytest.fixture(scope="module")
def hostsim_lib():
    return kc.build_hostsim()


@pytest.mark.parametrize("N,H,W,C,K", [(2, 6, 6, 32, 64), (1, 7, 5, 64, 32), (3, 4, 4, 32, 96), (1, 5, 6, 128, 64), (2, 4, 4, 128, 128)])
def test_winograd_fwd_bwd(hostsim_lib, N, H, W, C, K):
    kc.wino_case(hostsim_lib, "cpu", N, H, W, C, K, seed=C + K)


def test_weight_prep_equals_separate_transforms(hostsim_lib):
    kc.weight_prep_case(hostsim_lib, "cpu")


@pytest.mark.parametrize("N,H,W,C,K", [(2, 8, 8, 64, 64), (1, 7, 5, 64, 128), (3, 4, 4, 128, 64), (1, 14, 14, 64, 64), (2, 6, 9, 64, 64)])
def test_winograd_f4_fwd_bwd(hostsim_lib, N, H, W, C, K):
    """F(4x4, 3x3) (csrc/wino4.hip): forward (+ BN statistics, bias / ReLU epilogues), backward-data, the dual dy transform, backward-weight
    as slabs -- image sizes that are and are not multiples of 4."""
    assert hostsim_lib.wino_tile(N, H, W, C, K) == 4
    kc.wino_case(hostsim_lib, "cpu", N, H, W, C, K, seed=C + K + H, m=4)


def test_winograd_tile_choice(hostsim_lib):
    assert hostsim_lib.wino_tile(2, 3, 8, 64, 64) == 2          # H < 4
    assert hostsim_lib.wino_tile(2, 8, 8, 48, 64) == 2          # C not a multiple of 64: the slab GEMM does not take it
    assert hostsim_lib.wino_tile(32, 14, 14, 512, 512) == 4


@pytest.mark.parametrize("N,H,W,C,residual", [(2, 8, 8, 64, False), (1, 7, 9, 32, True), (2, 4, 4, 128, True), (1, 14, 14, 24, False)])
def test_bn_fused_into_winograd_input_transform(hostsim_lib, N, H, W, C, residual):
    kc.wino_bn_input_case(hostsim_lib, "cpu", N, H, W, C, residual, seed=C + H)


@pytest.mark.parametrize("N,H,W,C,residual,addend", [(2, 8, 8, 64, False, False), (1, 7, 9, 32, True, True), (2, 4, 4, 128, True, False),
                                                      (1, 14, 14, 24, False, True)])
def test_bn_backward_fused_into_winograd_transforms(hostsim_lib, N, H, W, C, residual, addend):
    kc.wino_bn_bwd_case(hostsim_lib, "cpu", N, H, W, C, residual, addend, seed=C + H)


# ---- the tile MOSAIC (round 4): 16 images share one map with single lines of zeros between them (H % 4 in {1, 2}, N % 16 == 0)
def test_winograd_mosaic_tile_count(hostsim_lib):
    assert hostsim_lib.wino_tiles(32, 14, 14, 4) == 480            # 2 x 15 x 15 = 450, rounded up to a multiple of 32 (plain: 512)
    assert hostsim_lib.wino_tiles(16, 6, 6, 4) == 64               # 7 x 7 = 49 -> 64 (plain: 16 x 4 = 64: no loss either)
    assert hostsim_lib.wino_tiles(16, 13, 13, 4) == 224            # 14 x 14 = 196 -> 224 (plain: 256)
    assert hostsim_lib.wino_tiles(8, 14, 14, 4) == 8 * 16          # N % 16 != 0: plain tiles
    assert hostsim_lib.wino_tiles(16, 28, 28, 4) == 16 * 49        # H % 4 == 0: plain tiles
    assert hostsim_lib.wino_tiles(16, 14, 14, 2) == 16 * 49        # F(2x2, 3x3): plain tiles


@pytest.mark.parametrize("N,H,C,K", [(16, 6, 64, 64), (16, 5, 64, 128), (32, 6, 64, 64)])
def test_winograd_f4_mosaic_fwd_bwd(hostsim_lib, N, H, C, K):
    """The whole F(4x4, 3x3) pipeline on mosaic tiles (forward + statistics + epilogues, backward-data, dual transform, backward-weight
    slabs) against torch conv2d: only the tile -> pixel mapping of the transform kernels differs from the plain form."""
    assert hostsim_lib.wino_tile(N, H, H, C, K) == 4 and hostsim_lib.wino_tiles(N, H, H, 4) < N * ((H + 3) // 4) ** 2 + 32
    kc.wino_case(hostsim_lib, "cpu", N, H, H, C, K, seed=C + K + H, m=4)


@pytest.mark.parametrize("N,H,C,residual,addend", [(16, 6, 64, False, False), (16, 5, 32, True, True)])
def test_bn_fusions_on_mosaic_tiles(hostsim_lib, N, H, C, residual, addend):
    kc.wino_bn_input_case(hostsim_lib, "cpu", N, H, H, C, residual, seed=C + H)
    kc.wino_bn_bwd_case(hostsim_lib, "cpu", N, H, H, C, residual, addend, seed=C + H)


def test_weight_gradient_transforms_of_several_layers_in_one_launch(hostsim_lib):
    kc.wino4_dw_multi_case(hostsim_lib, "cpu")
