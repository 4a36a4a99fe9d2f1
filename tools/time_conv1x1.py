"""1x1 / stride 1 convolutions on the GEMM kernels (csrc/gemm.hip) vs the implicit-GEMM kernels (HIFIHR_CONV1X1_GEMM=0), per direction.
usage: python tools/time_conv1x1.py            (run twice: once plain, once with HIFIHR_CONV1X1_GEMM=0)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hifihr_amd._lib import get_lib
lib = get_lib()
dev = "cuda"


def timeit(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


print("mode:", "implicit GEMM" if os.environ.get("HIFIHR_CONV1X1_GEMM") == "0" else "gemm.hip kernels")
# ResNet-18 layer4.0 projection; ResNet-50 bottleneck conv1 / conv3 shapes (layer 1 .. 4, layer-4 stride 1)
EFF = os.environ.get("EFFNET") == "1"            # EFFNET=1: EfficientNet-b3's 1x1 shapes at batch 48 (ragged channel counts); HIFIHR_GEMM_RAGGED=0 for the A/B
SHAPES_EFF = [(48, 112, 24, 144), (48, 56, 32, 192), (48, 28, 48, 288), (48, 14, 96, 576), (48, 14, 576, 136), (48, 14, 136, 816), (48, 14, 816, 136),
              (48, 7, 816, 232), (48, 7, 232, 1392), (48, 7, 1392, 232), (48, 7, 1392, 384), (48, 7, 384, 2304), (48, 7, 384, 1536)]
for N, H, C, K in SHAPES_EFF if EFF else [(32, 14, 256, 512), (32, 56, 64, 256), (32, 56, 256, 128), (32, 28, 128, 512), (32, 28, 512, 128), (32, 14, 256, 1024),
                   (32, 14, 1024, 256), (32, 14, 512, 2048), (32, 14, 2048, 512)]:
    M = N * H * H
    x = torch.randn(N, H, H, C, device=dev); w = torch.randn(K, 1, 1, C, device=dev) / C ** 0.5
    y = torch.empty(N, H, H, K, device=dev); dy = torch.randn(N, H, H, K, device=dev); dx = torch.empty_like(x)
    wt = w.reshape(K, C).t().contiguous()
    dw = torch.zeros(K, 1, 1, C, device=dev)
    stats = torch.zeros(lib.bn_stats_floats(K), device=dev)
    nws = lib.conv2d_wgrad_workspace_bytes(N, H, H, C, K, 1, 1, 1, 0)
    ws = torch.empty(max(nws // 4, 1), device=dev) if nws else None
    t_f = timeit(lambda: lib.conv2d_fwd(x, w, None, y, N, H, H, C, K, 1, 1, 1, 0))
    t_d = timeit(lambda: lib.conv2d_bwd_data_pre(dy, wt, dx, N, H, H, C, K, 1, 1, 1, 0))
    t_w = timeit(lambda: lib.conv2d_bwd_weight(x, dy, dw, N, H, H, C, K, 1, 1, 1, 0, ws=ws))
    fl = 2.0 * M * C * K
    print(f"N={N} {H}x{H} {C:4d}->{K:4d}: fwd {t_f:6.1f} us ({fl / t_f / 1e6:5.1f} TF)  dgrad {t_d:6.1f} us ({fl / t_d / 1e6:5.1f} TF)  "
          f"wgrad {t_w:6.1f} us ({fl / t_w / 1e6:5.1f} TF)   [{lib.conv2d_describe(N, H, H, C, K, 1, 1, 1, 0, 0)} / {lib.conv2d_describe(N, H, H, C, K, 1, 1, 1, 0, 2)}]")
