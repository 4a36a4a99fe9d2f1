"""MFMA implicit-GEMM conv kernel SOURCES on the hostsim emulator (exact MFMA emulation) vs torch conv2d."""
import pytest

import kernel_cases as kc


@pytest.fixture(scope="module")
def hostsim_lib():
    return kc.build_hostsim()


@pytest.mark.parametrize("N,H,W,C,K,R,stride,pad", [
    (2, 9, 7, 16, 64, 3, 1, 1),     # 3x3 s1, ragged M (126 rows), 64x64 tile path
    (1, 12, 12, 32, 128, 3, 2, 1),  # 3x3 s2 (dgrad divisibility path)
    (2, 8, 8, 64, 128, 1, 2, 0),    # 1x1 s2 downsample
    (1, 16, 16, 4, 64, 7, 2, 3),    # conv1 shape: C = 4 (NHWC4), Q = 196 not a multiple of 16
    (1, 6, 6, 48, 48, 3, 1, 0),     # K not a multiple of 64, no padding, bias
    (2, 5, 5, 24, 40, 1, 1, 0),     # EfficientNet-style 1x1 with C, K not multiples of 16 (generic gather, generic dgrad)
    (1, 7, 7, 136, 816, 1, 1, 0),   # 1x1 expand 136 -> 816
    (2, 9, 11, 4, 64, 3, 1, 1),     # VGG19 conv1_1 on NHWC4: its backward-data has 4 output channels (conv3x3_oc4_kernel, flipped taps)
    (1, 6, 5, 32, 4, 3, 1, 1),      # the same kernel in the forward direction
])
def test_conv_fwd_bwd(hostsim_lib, N, H, W, C, K, R, stride, pad):
    kc.conv_case(hostsim_lib, "cpu", N, H, W, C, K, R, stride, pad, seed=H, bias=(K == 48))


@pytest.mark.parametrize("N,H,W,C,K,R,stride,pad", [
    (2, 9, 7, 128, 192, 3, 1, 1),   # fwd 6 tiles x 36 chunks, dgrad 4 tiles x 54 chunks over 16 persistent workgroups: split tiles
    (3, 8, 8, 32, 128, 3, 1, 1),    # 6 tiles x 9 ... below the chunk minimum: must NOT take a workspace
    (4, 20, 20, 64, 64, 3, 2, 1),   # strided forward (one gather class) balanced, strided dgrad not
    (2, 16, 16, 64, 64, 3, 1, 1),   # 8 tiles on 16 workgroups: the in-phase split (one tile per "XCD": main part + one tail workgroup)
])
def test_conv_balanced_schedule(hostsim_lib, N, H, W, C, K, R, stride, pad):
    """hostsim reports 4 CUs -> 16 persistent workgroups (tests/hostsim/hip/hip_runtime.h)."""
    used = kc.conv_case(hostsim_lib, "cpu", N, H, W, C, K, R, stride, pad, seed=C + H)
    assert used == {192: 2, 128: 0, 64: 1 if stride == 2 else 2}[K]


def test_conv_balanced_schedule_bnstats(hostsim_lib):
    kc.conv_bnstats_case(hostsim_lib, "cpu", 2, 9, 7, 64, 192, 3, 1, 1, use_ws=True)


def test_image_to_nhwc4(hostsim_lib):
    kc.image_to_nhwc4_case(hostsim_lib, "cpu")


@pytest.mark.parametrize("N,H,W,C,K,R,stride", [(2, 8, 8, 128, 48, 1, 2), (1, 7, 7, 48, 48, 3, 1), (2, 6, 6, 48, 64, 3, 2), (1, 8, 8, 32, 48, 1, 4)])
def test_conv_bias_relu(hostsim_lib, N, H, W, C, K, R, stride):
    """LightEstimator layers: conv + bias + ReLU in one launch and the masked-gradient / bias-gradient kernel."""
    kc.conv_bias_relu_case(hostsim_lib, "cpu", N, H, W, C, K, R, stride, seed=C + K)


@pytest.mark.parametrize("N,H,W,C,K,R", [(1, 8, 8, 32, 128, 1), (1, 8, 8, 32, 64, 1), (2, 6, 6, 64, 256, 1), (1, 6, 6, 32, 128, 3)])
def test_conv_relu_without_bias(hostsim_lib, N, H, W, C, K, R):
    """conv2d_fwd(bias=None, act=1): clamped on every dispatch path (1x1 with K % 128 == 0 must not take the activation-less GEMM)."""
    kc.conv_relu_nobias_case(hostsim_lib, "cpu", N, H, W, C, K, R, 1, seed=K, pad=R // 2)


def test_conv_halo_bias_relu_epilogue(hostsim_lib):
    """VGG19's conv1_2 of the perceptual loss (64 -> 64, bias + ReLU): the halo kernel's epilogue adds the bias and clamps."""
    kc.conv_bias_relu_case(hostsim_lib, "cpu", 2, 10, 14, 64, 64, 3, 1, seed=5, pad=1)
    kc.conv_bias_relu_case(hostsim_lib, "cpu", 1, 5, 28, 64, 64, 3, 1, seed=6, pad=1)


@pytest.mark.parametrize("N,H,W", [(2, 10, 14), (1, 5, 28), (2, 24, 14), (1, 3, 42), (1, 6, 20), (2, 9, 31), (1, 4, 15)])
def test_conv_halo_layer1_shape(hostsim_lib, N, H, W):
    """3x3 / stride 1 / 64 -> 64 channels, W >= 14 (a ragged last column tile when W % 14 != 0): conv_halo_kernel (csrc/conv_halo.hip),
    forward and backward-data; the weight gradient's slab kernel only at W % 14 == 0.
    hostsim reports 4 CUs: shares of 5 rows (one 7-block tile), shares that cross a column-tile boundary, 8 + 4 row tiles, 3-row images."""
    assert hostsim_lib.conv2d_describe(N, H, W, 64, 64, 3, 3, 1, 1, 0) == "conv_halo_kernel"
    assert hostsim_lib.conv2d_describe(N, H, W, 64, 64, 3, 3, 1, 1, 2) == ("conv_halo_wgrad_kernel" if W % 14 == 0 else "conv_wgrad_kernel")
    kc.conv_case(hostsim_lib, "cpu", N, H, W, 64, 64, 3, 1, 1, seed=H + W)


@pytest.mark.parametrize("N,H,W", [(1, 8, 14), (2, 12, 14), (1, 4, 28), (2, 6, 28), (1, 30, 14), (3, 2, 14), (1, 8, 20), (2, 4, 30), (1, 10, 36)])
def test_conv_wino2_layer1_shape(hostsim_lib, N, H, W):
    """conv_wino2_kernel (csrc/conv_halo.hip): 8-row tiles (two MFMA row blocks), 4- and 2-row tiles (one row block, one column block per
    wave), 6-row tiles, shares that cross column tiles and images (hostsim reports 4 CUs), ragged last column tiles (W = 20, 30, 36: the
    512 x 512 perceptual loss is 36 x 14 + 8), forward + statistics and backward-data."""
    kc.conv_wino2_case(hostsim_lib, "cpu", N, H, W, seed=H + W)


@pytest.mark.parametrize("N,H,W,res", [(2, 10, 14, False), (3, 8, 28, True)])
def test_conv_c64_bwd_pair(hostsim_lib, N, H, W, res):
    """conv_c64_bwd_pair_kernel: both bodies in one launch, on the emulator's (small) CU count."""
    kc.conv_c64_bwd_pair_case(hostsim_lib, "cpu", N, H, W, seed=N + H, with_res=res)


def test_conv_wino2_bias_relu_epilogue(hostsim_lib):
    kc.conv_wino2_case(hostsim_lib, "cpu", 2, 10, 14, seed=3, bias_relu=True)


def test_conv_halo_bnstats(hostsim_lib):
    kc.conv_bnstats_case(hostsim_lib, "cpu", 2, 12, 14, 64, 64, 3, 1, 1)
    kc.conv_bnstats_case(hostsim_lib, "cpu", 1, 7, 19, 64, 64, 3, 1, 1)


@pytest.mark.parametrize("N,H,W", [(1, 56, 56), (2, 20, 28), (3, 6, 84)])
def test_conv_stem_kernel(hostsim_lib, N, H, W):
    """7x7 / stride 2 / pad 3 on NHWC4 input to 64 channels with the output width a multiple of 14: conv_stem_kernel (filter resident in
    LDS, halo staged once per tile); shares of 7 rows, partial tiles, 3-row outputs."""
    assert hostsim_lib.conv2d_describe(N, H, W, 4, 64, 7, 7, 2, 3, 0) == "conv_stem_kernel"
    kc.conv_case(hostsim_lib, "cpu", N, H, W, 4, 64, 7, 2, 3, seed=H + W)                                # four real channels
    kc.conv_case(hostsim_lib, "cpu", N, H, W, 4, 64, 7, 2, 3, seed=H + W + 1, zero_last_channel=True)    # the encoder's case: padding channel skipped


@pytest.mark.parametrize("N,H,W,C,K", [(2, 7, 5, 128, 256), (1, 9, 9, 64, 512), (3, 4, 4, 256, 128)])
def test_conv_1x1_runs_on_the_gemm_kernels(hostsim_lib, N, H, W, C, K):
    """1x1 / stride 1 convolutions (the stride-1 projection shortcut, bottleneck conv1 / conv3): forward and backward-data on
    bgemm_nt_rows_kernel, the weight gradient on bgemm_tn_kernel slabs, batch statistics through the extra pass."""
    assert hostsim_lib.conv2d_describe(N, H, W, C, K, 1, 1, 1, 0, 0) .startswith("bgemm_nt_rows_kernel<")
    assert hostsim_lib.conv2d_describe(N, H, W, C, K, 1, 1, 1, 0, 2).startswith("bgemm_")
    assert hostsim_lib.conv2d_wgrad_workspace_bytes(N, H, W, C, K, 1, 1, 1, 0) > 0
    kc.conv_case(hostsim_lib, "cpu", N, H, W, C, K, 1, 1, 0, seed=C + K)
    kc.conv_bnstats_case(hostsim_lib, "cpu", N, H, W, C, K, 1, 1, 0)


@pytest.mark.parametrize("N,H,W,C,K", [(1, 6, 6, 136, 232), (1, 5, 5, 144, 240), (2, 4, 4, 232, 1392), (1, 6, 6, 1392, 384), (2, 7, 5, 136, 816)])
def test_conv_1x1_ragged_channels_run_on_the_gemm_kernel(hostsim_lib, N, H, W, C, K):
    """EfficientNet's 1x1 convolutions (channel counts that are multiples of 4, not of 32 / 128): forward and backward-data on
    bgemm_nt_rows_kernel<RAGGED> -- operand segments past row N / column K come from a page of zeros, the epilogue stores and counts
    columns < N only -- with the batch-norm statistics from its epilogue.  (Taken where the 128-column tiles are >= 90 % full and the
    reduction has >= 128 channels: csrc/gemm.hip bgemm_nt_ragged_supported.)"""
    assert hostsim_lib.conv2d_describe(N, H, W, C, K, 1, 1, 1, 0, 0) .startswith("bgemm_nt_rows_kernel<")
    kc.conv_case(hostsim_lib, "cpu", N, H, W, C, K, 1, 1, 0, seed=C + K)
    kc.conv_bnstats_case(hostsim_lib, "cpu", N, H, W, C, K, 1, 1, 0)


def test_conv_stem_wgrad_three_channel_parameter(hostsim_lib):
    kc.stem_c3_wgrad_case(hostsim_lib, "cpu", N=1, H=56)


def test_conv_stem_bnstats(hostsim_lib):
    kc.conv_bnstats_case(hostsim_lib, "cpu", 2, 28, 56, 4, 64, 7, 2, 3)


@pytest.mark.parametrize("N,H,W,C,K,R,stride,pad", [
    (2, 12, 12, 64, 128, 3, 2, 1),    # ResNet layer2.0 conv1's shape class: 3x3 / stride 2, taps outside the image on every side
    (1, 9, 11, 32, 128, 3, 2, 1),     # odd sizes, one 32-channel block per tap
    (2, 10, 10, 64, 256, 1, 2, 0),    # the stride-2 1x1 projection shortcut, two column tiles
    (1, 7, 5, 96, 128, 3, 2, 1),      # three channel blocks per tap
])
def test_conv_rows_gather_forward(hostsim_lib, N, H, W, C, K, R, stride, pad):
    """bgemm_nt_rows_kernel<2>: the strided forward convolutions as the row-share GEMM with the patch gather in its loader waves (zero page
    for taps outside the image), forward + batch-norm statistics; the other directions of the same call stay on their kernels."""
    kc.conv_case(hostsim_lib, "cpu", N, H, W, C, K, R, stride, pad, seed=K + C)
    kc.conv_bnstats_case(hostsim_lib, "cpu", N, H, W, C, K, R, stride, pad, use_ws=False)


def test_strided_conv_and_downsample_conv_in_one_launch(hostsim_lib, monkeypatch):
    """hifihr_conv2d_fwd_bnstats_pair (bgemm_nt_rows_pair2_kernel) == the two separate launches; the emulator is told 16 compute units (the pair
    wants at least 8 workgroups per side)."""
    monkeypatch.setenv("HIFIHR_GEMM_CUS", "16")
    kc.conv_fwd_pair_case(hostsim_lib, "cpu", 2, 12, 12, 32, 128, 128, seed=3)
    kc.conv_fwd_pair_case(hostsim_lib, "cpu", 1, 10, 14, 64, 128, 256, seed=4)


def test_strided_dgrad_with_the_downsample_1x1_as_a_tap(hostsim_lib):
    """conv_igemm_kernel with ConvGeom::src2: the 1x1 / stride 2 convolution's data gradient inside parity class (0, 0) of the 3x3 / stride 2
    launch (even and odd image sizes, channel counts of one and of several 32-deep chunks per tap)."""
    kc.conv_dgrad_plus1x1_case(hostsim_lib, "cpu", 2, 12, 12, 32, 64, seed=5)
    kc.conv_dgrad_plus1x1_case(hostsim_lib, "cpu", 1, 9, 13, 16, 32, seed=6)
    kc.conv_dgrad_plus1x1_case(hostsim_lib, "cpu", 1, 8, 10, 64, 48, seed=7)


def test_strided_wgrad_with_the_downsample_1x1_in_the_same_launch(hostsim_lib):
    """conv_wgrad_kernel with dy2 / dw2: the column tiles of the 1x1 / stride 2 convolution's weight gradient behind the 3x3's (channel counts
    below, at and above one 64-column tile; odd sizes; several pixel splits)."""
    kc.conv_wgrad_plus1x1_case(hostsim_lib, "cpu", 2, 12, 12, 32, 64, seed=8)
    kc.conv_wgrad_plus1x1_case(hostsim_lib, "cpu", 1, 9, 13, 16, 40, seed=9)
    kc.conv_wgrad_plus1x1_case(hostsim_lib, "cpu", 1, 8, 10, 96, 48, seed=10)
