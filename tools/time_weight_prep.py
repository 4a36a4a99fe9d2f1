"""Times hifihr_weight_prep per job kind on the ResNet-18 layer shapes (B-independent)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hifihr_amd._lib import get_lib

lib = get_lib()
dev = "cuda"
wino = [(128, 128)] * 3 + [(256, 256)] * 3 + [(512, 256)] + [(512, 512)] * 3
direct = [(64, 64, 9)] * 4 + [(128, 64, 9), (128, 64, 1), (256, 128, 9), (256, 128, 1), (512, 256, 1)]


def table(kinds):
    jobs = []
    for K, C in wino:
        w = torch.randn(K, 3, 3, C, device=dev)
        for kind in kinds:
            if kind in (1, 2):
                jobs.append((w, torch.empty(16 * K * C, device=dev), K, C, 9, kind))
    if 0 in kinds:
        for K, C, RS in direct:
            w = torch.randn(K * RS * C, device=dev)
            jobs.append((w, torch.empty(K * RS * C, device=dev), K, C, RS, 0))
    return jobs


for kinds in ((0,), (1,), (2,), (0, 1, 2)):
    jobs = table(kinds)
    t = lib.prep_jobs(jobs, dev)
    for bpj in (32, 96, 256, 512):
        for _ in range(3):
            lib.weight_prep(t, len(jobs), bpj)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lib.weight_prep(t, len(jobs), bpj)
        e1.record()
        torch.cuda.synchronize()
        print(f"kinds {kinds} jobs {len(jobs)} blocks/job {bpj}: {e0.elapsed_time(e1) * 1e3 / 20:.1f} us")
