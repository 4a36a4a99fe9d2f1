import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hifihr_amd._lib import get_lib
lib = get_lib()
def timeit(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
B = 32
for H, C, K in ((28, 128, 128), (14, 256, 256), (14, 256, 512), (14, 512, 512)):
    for m in (2, 4):
        P = (m + 2) ** 2
        parts = lib.wino_wgrad_parts(B, H, H, C, K, m)
        dU = torch.randn(parts * P * K * C, device="cuda"); dw = torch.zeros(K, 3, 3, C, device="cuda")
        t = timeit(lambda: lib.wino_dw_transform_parts(dU, parts, dw, K, C, m))
        mb = (parts * P * K * C + 2 * 9 * K * C) * 4 / 1e6
        print(f"H={H} C={C} K={K} m={m}: parts {parts}, {t:6.1f} us, {mb:6.1f} MB -> {mb / t * 1e-3 * 1e3:5.2f} TB/s")
