"""torch.bmm (rocBLAS / hipBLASLt fp32) on the Winograd batched GEMM shapes of the ResNet-18 step vs this repo's kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hifihr_amd._lib import get_lib
lib = get_lib()
torch.backends.cuda.matmul.allow_tf32 = False


def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


B = 32
for H, C, K in ((28, 128, 128), (14, 256, 256), (14, 256, 512), (14, 512, 512)):
    T = B * (H // 2) * (H // 2)
    V = torch.randn(16, T, C, device="cuda"); U = torch.randn(16, K, C, device="cuda"); M = torch.empty(16, T, K, device="cuda")
    gf = 2.0 * 16 * T * C * K / 1e9
    t_bmm = timeit(lambda: torch.bmm(V, U.transpose(1, 2), out=M))
    nb = lib.wino_gemm_workspace_bytes(B, H, H, C, K)
    ws = torch.zeros(nb // 4 + 64, device="cuda") if nb else None
    t_own = timeit(lambda: lib.wino_gemm(V, U, M, B, H, H, C, K, ws=ws))
    # backward-weight shape: dU[16][K][C] = Y'^T V
    Y = torch.randn(16, T, K, device="cuda"); dU = torch.zeros(16, K, C, device="cuda")
    t_bmm_w = timeit(lambda: torch.bmm(Y.transpose(1, 2), V, out=dU))
    t_own_w = timeit(lambda: lib.wino_wgrad_gemm(V, Y, dU, B, H, H, C, K))
    print(f"T={T:5d} C={C:3d} K={K:3d} {gf:5.1f} GF | fwd: bmm {t_bmm:6.1f} us ({gf / t_bmm * 1e3:5.1f} TF)  own {t_own:6.1f} us ({gf / t_own * 1e3:5.1f} TF)"
          f" | wgrad: bmm {t_bmm_w:6.1f} us ({gf / t_bmm_w * 1e3:5.1f} TF)  own {t_own_w:6.1f} us ({gf / t_own_w * 1e3:5.1f} TF)")

print("forward GEMM layouts: NT = V[T,C] . U[K,C]^T (as stored today) vs NN = V[T,C] . Ut[C,K]")
for H, C, K in ((28, 128, 128), (14, 256, 256), (14, 256, 512), (14, 512, 256), (14, 512, 512)):
    T = B * (H // 2) * (H // 2)
    V = torch.randn(16, T, C, device="cuda"); U = torch.randn(16, K, C, device="cuda"); Ut = U.transpose(1, 2).contiguous(); M = torch.empty(16, T, K, device="cuda")
    gf = 2.0 * 16 * T * C * K / 1e9
    t_nt = timeit(lambda: torch.bmm(V, U.transpose(1, 2), out=M))
    t_nn = timeit(lambda: torch.bmm(V, Ut, out=M))
    print(f"T={T:5d} C={C:3d} K={K:3d}: NT {t_nt:6.1f} us ({gf / t_nt * 1e3:5.1f} TF)   NN {t_nn:6.1f} us ({gf / t_nn * 1e3:5.1f} TF)")
