#!/usr/bin/env python3
"""ONE Winograd product of the 512-channel layers, 36 x [450 x 512] . [512 x 512]^T, on split-bf16 MFMA inside the
production row-share kernel (tools/_probe/libhifihr_split_bf16.so, nt_rows_body<3>: same loader waves, same LDS image, same schedule; the
operands are (hi, lo) bf16 pairs occupying the bytes of the f32 values) against the f32-MFMA kernel on the same operands and float64.
Measurement only -- the headline path stays f32.  usage (GPU box, repo root): python3 tools/split_bf16_probe.py"""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from hifihr_amd._lib import HifihrLib
lib = HifihrLib(os.path.join(R, "tools", "_probe", "libhifihr_split_bf16.so"))
c = lib.c
vp, ci, cl = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
c.hifihr_probe_split_bf16.argtypes = [vp, vp, cl, vp]
c.hifihr_probe_bgemm_nt_bf16x3.argtypes = [vp, vp, vp, ci, ci, ci, ci, vp]
c.hifihr_bgemm_nt.argtypes = [vp, vp, vp, ci, ci, ci, ci, vp, ctypes.c_size_t, vp]
st = lambda: vp(torch.cuda.current_stream().cuda_stream)
p = lambda t: vp(t.data_ptr())


def us(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


torch.manual_seed(0)
for batch, M, N, K in ((36, 450, 512, 512), (36, 450, 256, 256), (36, 1568, 128, 128)):
    A = torch.randn(batch, M, K, device="cuda"); B = torch.randn(batch, N, K, device="cuda") * 0.05
    C32 = torch.empty(batch, M, N, device="cuda"); C16 = torch.empty(batch, M, N, device="cuda")
    As = torch.empty_like(A); Bs = torch.empty_like(B)                    # the split images: same bytes
    assert c.hifihr_probe_split_bf16(p(A), p(As), A.numel(), st()) == 0 and c.hifihr_probe_split_bf16(p(B), p(Bs), B.numel(), st()) == 0
    t_split = us(lambda: c.hifihr_probe_split_bf16(p(A), p(As), A.numel(), st()))
    f32 = lambda: c.hifihr_bgemm_nt(p(A), p(B), p(C32), M, N, K, batch, None, 0, st())
    b16 = lambda: c.hifihr_probe_bgemm_nt_bf16x3(p(As), p(Bs), p(C16), M, N, K, batch, st())
    assert f32() == 0 and b16() == 0
    t32, t16 = us(f32), us(b16)
    ref = torch.matmul(A[:4].double(), B[:4].double().transpose(1, 2))
    rms = float(ref.pow(2).mean().sqrt())
    e = lambda C: ((C[:4].double() - ref).abs().max().item() / rms, (C[:4].double() - ref).pow(2).mean().sqrt().item() / rms)
    flop = 2.0 * batch * M * N * K
    print(f"{batch} x [{M} x {K}] . [{N} x {K}]^T")
    print(f"  f32 MFMA (production kernel)   {t32:7.1f} us  {flop / t32 / 1e6:6.1f} TFLOP/s   max|err|/rms {e(C32)[0]:.2e}  rms err/rms {e(C32)[1]:.2e}")
    print(f"  bf16x3 in the same kernel      {t16:7.1f} us  {flop / t16 / 1e6:6.1f} TFLOP/s f32-equivalent   max|err|/rms {e(C16)[0]:.2e}  rms err/rms {e(C16)[1]:.2e}"
          f"   ({t32 / t16:.2f} x; splitting A as a pass of its own: {t_split:.1f} us -- in production the input transform's store)")
