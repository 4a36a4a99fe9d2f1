"""csrc/gemm.hip through the C ABI on the GPU: the Winograd GEMM shapes of the ResNet-18 step at B = 32 and ragged / small ones."""
import pytest
import torch

import kernel_cases as kc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from hifihr_amd._lib import get_lib
    return get_lib()


@pytest.mark.parametrize("M,N,K,batch", [(6272, 128, 128, 16), (1568, 256, 256, 16), (1568, 512, 256, 16), (1568, 256, 512, 16),
                                         (1568, 512, 512, 16), (98, 128, 64, 2), (130, 192, 32, 3), (40, 64, 96, 1)])
def test_bgemm_nt(lib, M, N, K, batch):
    kc.bgemm_case(lib, "cuda", M, N, K, batch, seed=M + N + K)


@pytest.mark.parametrize("M,N,T,batch", [(128, 128, 6272, 16), (256, 256, 1568, 16), (512, 256, 1568, 16), (512, 512, 1568, 16),
                                         (64, 128, 98, 2), (128, 64, 40, 1), (64, 64, 777, 1)])
def test_bgemm_tn(lib, M, N, T, batch):
    kc.bgemm_tn_case(lib, "cuda", M, N, T, batch, seed=M + T)


def test_bgemm_tn_is_bit_reproducible(lib):
    """No atomics: two launches give identical slabs."""
    gen = torch.Generator().manual_seed(5)
    M, N, T, batch = 256, 256, 1568, 16
    a = torch.randn(batch, T, M, generator=gen).cuda(); b = torch.randn(batch, T, N, generator=gen).cuda()
    parts = lib.bgemm_tn_parts(M, N, T, batch)
    c1 = torch.empty(parts, batch, M, N, device="cuda"); c2 = torch.empty_like(c1)
    lib.bgemm_tn(a, b, c1, M, N, T, batch, parts); lib.bgemm_tn(a, b, c2, M, N, T, batch, parts)
    assert torch.equal(c1, c2)


@pytest.mark.parametrize("M,N,K,batch", [(300, 256, 96, 3), (50, 128, 32, 5), (129, 384, 64, 2), (17, 128, 160, 9), (100000, 128, 32, 1),
                                         (6272, 128, 128, 7), (2352, 256, 256, 16)])
def test_bgemm_nt_row_shares(lib, M, N, K, batch):
    """bgemm_nt_rows_kernel on 256 workgroups: shares ending inside tiles (1..8 row blocks), crossing column-tile and problem boundaries,
    a B = 48-sized product and a very tall one."""
    assert lib.bgemm_describe(False, M, N, K) .startswith("bgemm_nt_rows_kernel<")
    assert kc.bgemm_case(lib, "cuda", M, N, K, batch, seed=M + K) == 0


def test_bgemm_nt_rows_is_bit_reproducible(lib):
    gen = torch.Generator().manual_seed(9)
    a = torch.randn(16, 1568, 512, generator=gen).cuda(); b = torch.randn(16, 512, 512, generator=gen).cuda()
    c1 = torch.empty(16, 1568, 512, device="cuda"); c2 = torch.empty_like(c1)
    lib.bgemm_nt(a, b, c1, 1568, 512, 512, 16); lib.bgemm_nt(a, b, c2, 1568, 512, 512, 16)
    assert torch.equal(c1, c2)


@pytest.mark.parametrize("M,N,T,batch", [(512, 512, 512, 36), (512, 256, 512, 36), (256, 512, 768, 36), (320, 384, 96, 40), (128, 128, 64, 300)])
def test_bgemm_tn_row_shares(lib, M, N, T, batch):
    """bgemm_tn_rows_kernel on 256 workgroups (the F(4x4, 3x3) backward-weight shapes of layer 4 and shapes whose shares end inside tiles:
    64 / 32 / 16-row tail tiles, column-tile and problem boundaries): complete products in one slab, bit-reproducible."""
    assert lib.bgemm_describe(True, M, N, T, batch) == "bgemm_tn_rows_kernel"
    assert kc.bgemm_tn_case(lib, "cuda", M, N, T, batch, seed=M + T) == 1
    gen = torch.Generator().manual_seed(3)
    a = torch.randn(batch, T, M, generator=gen).cuda(); b = torch.randn(batch, T, N, generator=gen).cuda()
    c1 = torch.empty(1, batch, M, N, device="cuda"); c2 = torch.full_like(c1, 7.0)
    lib.bgemm_tn(a, b, c1, M, N, T, batch, 1); lib.bgemm_tn(a, b, c2, M, N, T, batch, 1)
    assert torch.equal(c1, c2)
