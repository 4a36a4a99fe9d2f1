"""What one dependent kernel edge costs inside a replayed hipGraph on this runtime: N trivial kernels captured back to back."""
import os, sys, time
import torch
dev = torch.device("cuda")
s = torch.cuda.Stream()
torch.cuda.set_stream(s)
x = torch.zeros(256, device=dev)
big = torch.zeros(1 << 22, device=dev)
def body(n):
    for _ in range(n):
        x.add_(1.0)
for n in (200,):
    body(10); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        body(n)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 20
    for _ in range(R): g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R
    print(f"{os.environ.get('TAG','default')}: {n} trivial kernels per replay: {dt*1e6/n:.2f} us per kernel")
