"""csrc/gemm.hip (hand-written batched f32 GEMM, 16x16x4 MFMA) on the Winograd GEMM shapes of the ResNet-18 step (B = 32):
tile / split variants vs round 1's gather kernels (HIFIHR_BGEMM=0) and vs the vendor library (torch.bmm), which is only the
yardstick here -- the product path no longer calls it."""
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


def setenv(k, v):
    if v is None: os.environ.pop(k, None)
    else: os.environ[k] = str(v)


B = int(os.environ.get("B", 32))
shapes = ((28, 128, 128), (14, 256, 256), (14, 256, 512), (14, 512, 256), (14, 512, 512))
print("== NT: M[16][T][K] = V[16][T][C] . U[16][K][C]^T")
for H, C, K in shapes:
    T = B * (H // 2) * (H // 2)
    V = torch.randn(16, T, C, device="cuda"); U = torch.randn(16, K, C, device="cuda"); M = torch.empty(16, T, K, device="cuda")
    gf = 2.0 * 16 * T * C * K / 1e9
    ref = torch.bmm(V, U.transpose(1, 2))
    t_bmm = timeit(lambda: torch.bmm(V, U.transpose(1, 2), out=M))
    setenv("HIFIHR_BGEMM", 0)
    nb = lib.wino_gemm_workspace_bytes(B, H, H, C, K)
    ws = torch.zeros(nb // 4 + 64, device="cuda") if nb else None
    t_old = timeit(lambda: lib.wino_gemm(V, U, M, B, H, H, C, K, ws=ws))
    setenv("HIFIHR_BGEMM", None)
    line = f"T={T:5d} C={C:3d} K={K:3d} {gf:5.1f} GF | bmm {t_bmm:6.1f} us ({gf / t_bmm * 1e3:5.1f} TF) | r1 kernel {t_old:6.1f} |"
    for tile, ws_, sk in ((128128, 4, 1), (128128, 2, 1), (128128, 4, 0), (128128, 2, 0), (128128, 0, 0), (64064, 0, 0)):
        setenv("HIFIHR_GEMM_NT_TILE", tile); setenv("HIFIHR_GEMM_WS", ws_); setenv("HIFIHR_GEMM_SK", sk)
        nbk = lib.wino_gemm_workspace_bytes(B, H, H, C, K)
        wsk = torch.zeros(nbk // 4 + 64, device="cuda") if nbk else None
        if sk and not nbk:
            continue
        M.fill_(7.0)
        lib.wino_gemm(V, U, M, B, H, H, C, K, ws=wsk)
        err = float((M - ref).abs().max() / ref.abs().max())
        t = timeit(lambda: lib.wino_gemm(V, U, M, B, H, H, C, K, ws=wsk))
        clean = "" if wsk is None or float(wsk.abs().max()) == 0.0 else " WS-DIRTY"
        line += f" {tile // 1000}x{tile % 1000}/ws{ws_}/{'sk' if sk else 'tile'}: {t:6.1f} us ({gf / t * 1e3:5.1f} TF, err {err:.1e}{clean})"
    setenv("HIFIHR_GEMM_NT_TILE", None); setenv("HIFIHR_GEMM_WS", None); setenv("HIFIHR_GEMM_SK", None)
    M.fill_(7.0)                                    # the default path: bgemm_nt_rows_kernel where N % 128 == 0
    lib.wino_gemm(V, U, M, B, H, H, C, K)
    err = float((M - ref).abs().max() / ref.abs().max())
    t = timeit(lambda: lib.wino_gemm(V, U, M, B, H, H, C, K))
    line += f" || default = {lib.bgemm_describe(False, T, K, C)}: {t:6.1f} us ({gf / t * 1e3:5.1f} TF, err {err:.1e})"
    print(line, flush=True)

print("== TN: dU[16][K][C] = Y'[16][T][K]^T . V[16][T][C]   (slabs, summed by the dw transform)")
for H, C, K in shapes:
    T = B * (H // 2) * (H // 2)
    V = torch.randn(16, T, C, device="cuda"); Y = torch.randn(16, T, K, device="cuda"); dU = torch.zeros(16, K, C, device="cuda")
    dw = torch.zeros(K, 3, 3, C, device="cuda")
    gf = 2.0 * 16 * T * C * K / 1e9
    ref = torch.bmm(Y.transpose(1, 2), V)
    t_bmm = timeit(lambda: torch.bmm(Y.transpose(1, 2), V, out=dU))
    dU.zero_()
    def old():
        lib.wino_wgrad_gemm(V, Y, dU, B, H, H, C, K); lib.wino_dw_transform(dU, dw, K, C, clear=True)
    t_old = timeit(old)
    line = f"T={T:5d} C={C:3d} K={K:3d} {gf:5.1f} GF | bmm {t_bmm:6.1f} us ({gf / t_bmm * 1e3:5.1f} TF) | r1 gemm+dw {t_old:6.1f} |"
    for tile, ws_ in ((128128, 4), (64064, 0)):
        setenv("HIFIHR_GEMM_TN_TILE", tile); setenv("HIFIHR_GEMM_WS", ws_)
        for parts_req in (None,):
            setenv("HIFIHR_GEMM_TN_PARTS", parts_req)
            parts = lib.wino_wgrad_parts(B, H, H, C, K)
            if parts_req is not None and parts != parts_req:
                continue
            try:
                dUp = torch.full((parts, 16, K, C), 7.0, device="cuda")
                lib.wino_wgrad_gemm_parts(V, Y, dUp, B, H, H, C, K, parts)
            except Exception as e:       # a forced part count the chunk split cannot produce
                continue
            err = float((dUp.sum(0) - ref).abs().max() / ref.abs().max())
            t_g = timeit(lambda: lib.wino_wgrad_gemm_parts(V, Y, dUp, B, H, H, C, K, parts))
            t_d = timeit(lambda: lib.wino_dw_transform_parts(dUp, parts, dw, K, C))
            line += f" {tile // 1000}x{tile % 1000}/ws{ws_}/p{parts}{'*' if parts_req is None else ''}: {t_g:5.1f}+{t_d:4.1f} ({gf / t_g * 1e3:5.1f} TF, {err:.0e})"
    setenv("HIFIHR_GEMM_TN_TILE", None); setenv("HIFIHR_GEMM_TN_PARTS", None); setenv("HIFIHR_GEMM_WS", None)
    print(line, flush=True)
