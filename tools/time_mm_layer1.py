"""Reference point for the direct-convolution main loop: the layer-1 implicit GEMM (M = 32*56*56, K = 576, N = 64) as a PLAIN GEMM on the
vendor library (the im2col matrix is materialised up front and not timed), vs conv_igemm_kernel on the same layer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hifihr_amd._lib import get_lib
lib = get_lib()


def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for (N, H, C, K) in ((32, 56, 64, 64), (32, 28, 128, 128)):
    M, Kd = N * H * H, 9 * C
    A = torch.randn(M, Kd, device="cuda"); W = torch.randn(K, Kd, device="cuda"); Y = torch.empty(M, K, device="cuda")
    gf = 2.0 * M * Kd * K / 1e9
    t = timeit(lambda: torch.mm(A, W.t(), out=Y))
    x = torch.randn(N, H, H, C, device="cuda"); w = torch.randn(K, 3, 3, C, device="cuda"); y = torch.empty(N, H, H, K, device="cuda")
    nb = lib.conv2d_workspace_bytes(N, H, H, C, K, 3, 3, 1, 1, False)
    ws = torch.zeros(nb // 4 + 64, device="cuda") if nb else None
    t2 = timeit(lambda: lib.conv2d_fwd(x, w, None, y, N, H, H, C, K, 3, 3, 1, 1, ws=ws))
    print(f"M={M} K={Kd} N={K} {gf:.1f} GF: library plain GEMM {t:6.1f} us ({gf / t * 1e3:5.1f} TF) | conv_igemm (gathers the taps itself) {t2:6.1f} us ({gf / t2 * 1e3:5.1f} TF)")
