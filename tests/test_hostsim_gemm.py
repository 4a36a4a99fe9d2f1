"""Batched f32 GEMM kernel SOURCES (csrc/gemm.hip: 16x16x4 MFMA, direct-to-LDS operand images) on the hostsim emulator vs torch."""
import pytest

import kernel_cases as kc


@pytest.fixture(scope="module")
def hostsim_lib():
    return kc.build_hostsim()


@pytest.mark.parametrize("M,N,K,batch", [(128, 128, 32, 1), (98, 128, 64, 2), (200, 64, 96, 1), (130, 192, 32, 2), (40, 256, 64, 1)])
def test_bgemm_nt(hostsim_lib, M, N, K, batch):
    kc.bgemm_case(hostsim_lib, "cpu", M, N, K, batch, seed=M + N)


@pytest.mark.parametrize("M,N,T,batch", [(128, 128, 64, 1), (64, 128, 98, 2), (128, 64, 40, 1), (64, 64, 33, 2), (256, 128, 320, 1), (64, 64, 777, 1)])
def test_bgemm_tn(hostsim_lib, M, N, T, batch):
    kc.bgemm_tn_case(hostsim_lib, "cpu", M, N, T, batch, seed=M + T)


@pytest.mark.parametrize("ws", [0, 1, 2, 4])
def test_bgemm_wave_specialised_forms(hostsim_lib, ws, monkeypatch):
    """128x128 tiles forced (the shape heuristic picks them only for long reductions): the 4-wave kernel (ws = 0) and the
    wave-specialised one with 1 / 2 / 4 loader waves, ragged M, ragged T, several slabs."""
    monkeypatch.setenv("HIFIHR_GEMM_NT_TILE", "128128")
    monkeypatch.setenv("HIFIHR_GEMM_TN_TILE", "128128")
    monkeypatch.setenv("HIFIHR_GEMM_WS", str(ws))
    kc.bgemm_case(hostsim_lib, "cpu", 200, 128, 160, 2, seed=ws)
    kc.bgemm_case(hostsim_lib, "cpu", 128, 256, 32, 1, seed=ws + 10)
    kc.bgemm_tn_case(hostsim_lib, "cpu", 128, 128, 300, 2, seed=ws + 20)
    monkeypatch.setenv("HIFIHR_GEMM_TN_PARTS", "3")
    assert kc.bgemm_tn_case(hostsim_lib, "cpu", 128, 256, 32 * 9 + 5, 1, seed=ws + 30) == 3


@pytest.mark.parametrize("ws", [2, 4])
def test_bgemm_persistent_balanced_form(hostsim_lib, ws, monkeypatch):
    """128x128 tiles + a workspace: the persistent kernel (4 workgroups on the emulator's 4 CUs), stream-K shares that split tiles
    between neighbouring workgroups (slab + flag hand-off), ragged M, several problems per batch; the workspace comes back clean."""
    monkeypatch.setenv("HIFIHR_GEMM_NT_TILE", "128128")
    monkeypatch.setenv("HIFIHR_GEMM_WS", str(ws))
    assert kc.bgemm_case(hostsim_lib, "cpu", 300, 128, 96, 5, seed=ws) > 0          # 15 tiles x 3 chunks over 4 workgroups
    assert kc.bgemm_case(hostsim_lib, "cpu", 128, 256, 160, 6, seed=ws + 1) > 0     # 12 tiles x 5 chunks
    assert kc.bgemm_case(hostsim_lib, "cpu", 130, 128, 64, 7, seed=ws + 2) > 0      # 14 tiles x 2 chunks
    assert kc.bgemm_case(hostsim_lib, "cpu", 300, 128, 96, 3, seed=ws) == 0         # 9 tiles < 3 rounds: one workgroup per tile
    monkeypatch.setenv("HIFIHR_GEMM_SK", "0")
    assert kc.bgemm_case(hostsim_lib, "cpu", 300, 128, 96, 5, seed=ws) == 0         # switched off


@pytest.mark.parametrize("M,N,K,batch", [(300, 256, 96, 3), (50, 128, 32, 5), (129, 384, 64, 2), (17, 128, 160, 9), (1000, 128, 32, 1)])
def test_bgemm_nt_row_shares(hostsim_lib, M, N, K, batch):
    """bgemm_nt_rows_kernel (N % 128 == 0, the default NT path): persistent workgroups (4 on the emulator) over equal shares of the
    (problem, column tile, row) space -- shares that end inside a tile (short tiles with 1..8 row blocks), that cross column-tile and
    problem boundaries, and chunk streams that run across tile boundaries."""
    assert hostsim_lib.bgemm_describe(False, M, N, K) .startswith("bgemm_nt_rows_kernel<")
    assert kc.bgemm_case(hostsim_lib, "cpu", M, N, K, batch, seed=M + K) == 0        # no workspace


@pytest.mark.parametrize("M,N,T,batch", [(128, 128, 64, 4), (192, 256, 96, 2), (64, 128, 128, 8), (512, 128, 64, 1), (320, 128, 96, 3), (448, 384, 64, 1)])
def test_bgemm_tn_row_shares(hostsim_lib, M, N, T, batch):
    """bgemm_tn_rows_kernel (N % 128 == 0, M % 64 == 0, T % 32 == 0, at least eight 16-row blocks per CU; hostsim reports 4 CUs): shares that end
    inside a 128-row tile, tails cut into 64 / 32 / 16-row tiles, tiles that cross problem boundaries, complete products in ONE slab."""
    assert hostsim_lib.bgemm_describe(True, M, N, T, batch) == "bgemm_tn_rows_kernel"
    assert kc.bgemm_tn_case(hostsim_lib, "cpu", M, N, T, batch, seed=M + T) == 1


@pytest.mark.parametrize("M,N,T,batch,cus", [(256, 128, 64, 20, 16), (256, 256, 64, 18, 16), (128, 256, 96, 9, 8)])
def test_bgemm_tn_xcd_coherent_schedule(hostsim_lib, monkeypatch, M, N, T, batch, cus):
    """bgemm_tn_rows_kernel on the XCD-coherent schedule (BgemmArgs::co_r: rounds in which the workgroups of one XCD split the tiles of the
    same problem(s), then a contiguous tail over the problems behind the last full round) -- the emulator is told `cus` compute units so that
    the workgroup count is a multiple of 8: whole-tile slices, problem boundaries inside a round, a tail of short tiles, and the contiguous
    schedule (HIFIHR_GEMM_TN_COHERENT=0) giving the SAME bits (every output element is one complete reduction either way)."""
    import torch
    monkeypatch.setenv("HIFIHR_GEMM_CUS", str(cus))
    assert hostsim_lib.bgemm_describe(True, M, N, T, batch) == "bgemm_tn_rows_kernel"
    assert kc.bgemm_tn_case(hostsim_lib, "cpu", M, N, T, batch, seed=M + T + batch) == 1
    gen = torch.Generator().manual_seed(5)
    a = torch.randn(batch, T, M, generator=gen); b = torch.randn(batch, T, N, generator=gen)
    c1 = torch.full((1, batch, M, N), 7.0); c0 = torch.full((1, batch, M, N), 7.0)
    hostsim_lib.bgemm_tn(a, b, c1, M, N, T, batch, 1)
    monkeypatch.setenv("HIFIHR_GEMM_CUS", "4")               # 4 workgroups: never coherent
    hostsim_lib.bgemm_tn(a, b, c0, M, N, T, batch, 1)
    assert torch.equal(c0, c1)


def _fma32(a, b, d):
    """round-to-nearest float32 of a * b + d for float32 arrays (the emulator's MFMA is a chain of fmaf): the product of two 24-bit
    significands is exact in the 64-bit significand of a long double, the sum is rounded once to it and once to float32 -- a double rounding
    that differs from fmaf with probability ~2^-40 per operation on random operands"""
    import numpy as np
    return (a.astype(np.longdouble) * b.astype(np.longdouble) + d.astype(np.longdouble)).astype(np.float32)


def test_row_share_kernels_keep_the_summation_order(hostsim_lib):
    """The row-share kernels finish a tile block by block (round 6: the last chunk runs row block by row block so that a block's stores go out
    between the MFMAs of the blocks behind it; the TN body also takes the live k-step of the zero-row tail from LDS inside that loop).  Whatever
    the order of the MFMAs across row blocks, every accumulator must see its k-steps in the SAME order as in the plain loop (TN: t ascending;
    NT: chunks ascending, inside a chunk the fixed order of the fragment layout): on the emulator (v_mfma = four chained fmaf) the result then
    equals, bit for bit, a plain chain of fmaf in that order per output element -- for tiles of every height (1 .. 8 row blocks), shares that
    cross problems, and a TN product behind a tile mosaic whose last rows are zero (the skipped k-steps would add exact zeros)."""
    import numpy as np
    import torch
    assert np.finfo(np.longdouble).nmant >= 63, "needs an x87 long double"
    gen = torch.Generator().manual_seed(11)
    # NT: C[m][n] = sum_k A[m][k] B[n][k]
    for M, N, K, batch in ((300, 128, 96, 3), (50, 256, 64, 2), (129, 128, 32, 1)):
        a = torch.randn(batch, M, K, generator=gen); b = torch.randn(batch, N, K, generator=gen)
        assert hostsim_lib.bgemm_describe(False, M, N, K).startswith("bgemm_nt_rows_kernel<")
        c = torch.full((batch, M, N), 7.0)
        hostsim_lib.bgemm_nt(a, b, c, M, N, K, batch)
        an, bn = a.numpy(), b.numpy()
        ref = np.zeros((batch, M, N), np.float32)
        # (the NT body's order inside a 32-deep chunk: half h, component kc of the lanes' 16-byte fragments, lane group g -> k = 16 h + 4 g + kc)
        for k in [32 * ch + 16 * h + 4 * g + kc for ch in range(K // 32) for h in range(2) for kc in range(4) for g in range(4)]:
            ref = _fma32(an[:, :, k, None], bn[:, None, :, k], ref)
        assert np.array_equal(c.numpy(), ref), (M, N, K, batch, float(np.abs(c.numpy() - ref).max()))
    # TN: C[m][n] = sum_t A[t][m] B[t][n]
    for M, N, T, batch in ((128, 128, 96, 4), (192, 256, 96, 2), (320, 128, 64, 3)):
        a = torch.randn(batch, T, M, generator=gen); b = torch.randn(batch, T, N, generator=gen)
        assert hostsim_lib.bgemm_describe(True, M, N, T, batch) == "bgemm_tn_rows_kernel"
        c = torch.full((1, batch, M, N), 7.0)
        hostsim_lib.bgemm_tn(a, b, c, M, N, T, batch, 1)
        an, bn = a.numpy(), b.numpy()
        ref = np.zeros((batch, M, N), np.float32)
        for t in range(T):
            ref = _fma32(an[:, t, :, None], bn[:, t, None, :], ref)
        assert np.array_equal(c.numpy()[0], ref), (M, N, T, batch, float(np.abs(c.numpy()[0] - ref).max()))
    # TN behind a tile mosaic: 16 images of 13 x 13 share 196 F(4x4) tiles in 224 allocated rows; rows 196 .. 223 of both operands are zero
    # and their k-steps are skipped (BgemmArgs::k_valid) -- 6 full chunks + ONE live k-step: the fused form of the last full chunk
    Nn, H, C, K = 16, 13, 128, 64
    T4, Tr = hostsim_lib.wino_tiles(Nn, H, H, 4), hostsim_lib.wino_tiles_computed(Nn, H, H, 4)
    assert (T4, Tr) == (224, 196) and hostsim_lib.bgemm_describe(True, K, C, T4, 36) == "bgemm_tn_rows_kernel"
    parts = hostsim_lib.wino_wgrad_parts(Nn, H, H, C, K, 4)
    assert parts == 1
    V = torch.randn(36, T4, C, generator=gen); Y = torch.randn(36, T4, K, generator=gen)
    V[:, Tr:] = 0; Y[:, Tr:] = 0
    dU = torch.full((parts, 36, K, C), 7.0)
    hostsim_lib.wino_wgrad_gemm_parts(V, Y, dU, Nn, H, H, C, K, parts, m=4)
    vn, yn = V.numpy(), Y.numpy()
    ref = np.zeros((36, K, C), np.float32)
    for t in range(Tr):
        ref = _fma32(yn[:, t, :, None], vn[:, t, None, :], ref)
    assert np.array_equal(dU.numpy()[0], ref), float(np.abs(dU.numpy()[0] - ref).max())
