"""Kernel cases of tests/kernel_cases.py on the AddressSanitizer build of the HIP emulator (tests/hostsim: `make asan`).
Run through tools/asan_hostsim.sh (libasan must be preloaded before python starts).  usage: asan_hostsim.py <group> ..."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import kernel_cases as kc  # noqa: E402
from hifihr_amd._lib import HifihrLib  # noqa: E402
from hifihr_amd.mano_tables import synthetic_mano_tables  # noqa: E402

lib = HifihrLib(os.path.join(R, "tests", "hostsim", "libhifihr_hostsim_asan.so"))


def conv():
    for (N, H, W) in [(1, 8, 14), (2, 12, 14), (1, 4, 28), (3, 2, 14), (1, 8, 20), (2, 4, 30)]:         # (20, 30: a ragged last column tile)
        kc.conv_wino2_case(lib, "cpu", N, H, W, seed=H + W)
    kc.conv_wino2_case(lib, "cpu", 2, 10, 14, seed=3, bias_relu=True)
    for (N, H, W) in [(2, 10, 14), (1, 6, 20), (1, 4, 15)]:
        kc.conv_case(lib, "cpu", N, H, W, 64, 64, 3, 1, 1, seed=H + W)                     # conv_halo_kernel / conv_halo_wgrad_kernel
    kc.conv_case(lib, "cpu", 1, 20, 28, 4, 64, 7, 2, 3, seed=1)                              # stem kernels
    kc.conv_case(lib, "cpu", 2, 9, 7, 16, 64, 3, 1, 1, seed=2)                               # implicit GEMM
    kc.conv_case(lib, "cpu", 2, 7, 5, 128, 256, 1, 1, 0, seed=3)                             # 1x1 on the GEMM kernels
    kc.conv_case(lib, "cpu", 1, 6, 6, 136, 232, 1, 1, 0, seed=4)                             # ... with ragged N and K (zero-page segments)
    kc.conv_bnstats_case(lib, "cpu", 1, 6, 6, 1392, 384, 1, 1, 0)
    kc.conv_bnstats_case(lib, "cpu", 2, 12, 14, 64, 64, 3, 1, 1)


def render():
    t = synthetic_mano_tables(0)
    for (s, aa) in [(32, 3), (36, 2), (20, 3)]:
        kc.render_case(lib, t, "cpu", 2, 1, s, aa)
    kc.render_uv_case(lib, t, "cpu", 2, 1, 32, 3)


def wino():
    kc.wino_bn_input_case(lib, "cpu", 2, 12, 12, 64, False, seed=1)
    kc.wino_bn_input_case(lib, "cpu", 1, 9, 7, 128, True, seed=2)
    kc.wino_bn_bwd_case(lib, "cpu", 2, 12, 12, 64, False, False, seed=3)
    kc.wino_bn_bwd_case(lib, "cpu", 1, 9, 7, 64, True, True, seed=4)


def round5():
    """Kernels written or re-indexed in round 5: the layer-1 pair launch, the flattening pool, SSIM's 48-column frame (vector and scalar
    staging, ragged image sizes), the four-pixel warp kernel through the batch entry, Adam with the counter on the device."""
    import numpy as np
    import torch
    for (N, H, W, res) in [(2, 10, 14, False), (3, 8, 28, True), (1, 6, 14, True)]:
        kc.conv_c64_bwd_pair_case(lib, "cpu", N, H, W, seed=N + H, with_res=res)
    for ksp in ((3, 2, 1), (3, 1, 1), (2, 2, 0)):
        kc.maxpool_case(lib, "cpu", 2, 12, 10, 8, seed=ksp[0], ties=True, ksp=ksp)
    gen = torch.Generator().manual_seed(5)
    for (H, W) in [(64, 64), (40, 36), (33, 30), (70, 44)]:          # W % 4 == 0: vector staging; 30: the scalar form; partial tiles
        a = torch.rand(2, 3, H, W, generator=gen).numpy(); b = torch.rand(2, 3, H, W, generator=gen).numpy()
        kc.ssim_case(lib, "cpu", a, b, None, None)
    kc.freihand_batch_case(lib, "cpu", seed=1)
    kc.freihand_batch_case(lib, "cpu", seed=2, B=3, n=5, res=32, J=21, V=50)
    kc.adam_case(lib, "cpu", n=4099, wd=0.01, steps=4)


def round6():
    """Kernels written or re-indexed in round 6: the multi-layer weight-gradient transform and slab sum, the stem's batch-norm reduction over
    the pooled grid (incl. the tap-gather path of a near-zero scale), the TN row-share kernel's zero-row tail and XCD-coherent schedule, the
    light split with a NaN colour."""
    import torch
    kc.wino4_dw_multi_case(lib, "cpu")
    for (N, H, W, res) in [(2, 10, 14, False), (1, 6, 14, True)]:
        kc.conv_c64_bwd_pair_case(lib, "cpu", N, H, W, seed=N + H, with_res=res)      # (+ the _slabs form and the multi-layer slab sum)
    for (N, H, W, C) in [(2, 9, 11, 8), (3, 12, 12, 64), (1, 7, 8, 16)]:
        kc.bn_relu_maxpool_case(lib, "cpu", N, H, W, C, seed=H + C)
    for m in (4,):                                                               # F(4x4): mosaic layers (zero rows behind the last mosaic)
        kc.wino_case(lib, "cpu", 16, 14, 14, 64, 128, seed=3, m=m)
    os.environ["HIFIHR_GEMM_CUS"] = "16"
    try:
        kc.bgemm_tn_case(lib, "cpu", 256, 128, 64, 20, seed=7)                    # coherent rounds + a tail of short tiles
        kc.bgemm_tn_case(lib, "cpu", 128, 256, 96, 9, seed=8)
    finally:
        del os.environ["HIFIHR_GEMM_CUS"]
    kc.light_split_case(lib, "cpu", B=5)
    # the row-share GEMM bodies after this round's changes (division-free walkers, head chunk, the row-block-wise last chunk with the stores
    # between the MFMAs, the TN body's fused mosaic tail): tiles of several heights, shares across problems, 16 images of 13 x 13 on a mosaic
    kc.bgemm_case(lib, "cpu", 300, 128, 96, 3, seed=1)
    kc.bgemm_case(lib, "cpu", 50, 256, 64, 2, seed=2)
    kc.bgemm_tn_case(lib, "cpu", 192, 256, 96, 2, seed=3)
    kc.wino_case(lib, "cpu", 16, 13, 13, 64, 128, seed=4, m=4)
    # the strided blocks' backward with the downsample 1x1 convolution folded in (conv_igemm_kernel's second source, conv_wgrad_kernel's extra tiles)
    kc.conv_dgrad_plus1x1_case(lib, "cpu", 1, 9, 13, 16, 32, seed=6)
    kc.conv_wgrad_plus1x1_case(lib, "cpu", 1, 9, 13, 16, 40, seed=9)
    os.environ["HIFIHR_GEMM_CUS"] = "16"                                          # (the pair wants at least 8 workgroups per side)
    try:
        kc.conv_fwd_pair_case(lib, "cpu", 2, 12, 12, 32, 128, 128, seed=3)
    finally:
        del os.environ["HIFIHR_GEMM_CUS"]


GROUPS = {"conv": conv, "render": render, "wino": wino, "round5": round5, "round6": round6}
for g in (sys.argv[1:] or list(GROUPS)):
    GROUPS[g]()
    print(f"asan: {g} clean", flush=True)
