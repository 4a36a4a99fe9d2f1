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
