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


GROUPS = {"conv": conv, "render": render, "wino": wino}
for g in (sys.argv[1:] or list(GROUPS)):
    GROUPS[g]()
    print(f"asan: {g} clean", flush=True)
