#!/usr/bin/env python3
"""Which parts of the training step are bit-reproducible?  forward + backward (no optimizer step) from IDENTICAL weights, several times
eagerly and several times as a replayed graph: the loss terms and the gradient of every parameter are compared bit for bit.
Lists the parameters whose gradients differ between two eager runs (float-atomic ordering) and between eager and graph.
usage: determinism_probe.py [B] [runs]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from hifihr_amd import options, synth
from hifihr_amd.losses import LossFunction
from hifihr_amd.mano_tables import synthetic_mano_tables
from hifihr_amd.models import Model
from hifihr_amd.optim import FlatParams, FusedAdam
from hifihr_amd.traineval import data_dic, forward_backward, _prime_for_capture
from test_gpu_e2e import graded_images

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
RUNS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.cuda.set_stream(torch.cuda.Stream())
dev = torch.device("cuda")
tables = synthetic_mano_tables(0)
args = options.baseline_config2_args(train_batch=B)
torch.manual_seed(0)
model = Model(True, dev, False, "mano", False, "res18", mano_tables=tables).to(dev).train()
sample = synth.make_batch(model.hand_layer.handle, model.renderer_p3d, B, first_index=0, device=dev)
sample["trans_images"] = graded_images(sample["trans_images"])
ex = data_dic(sample, "FreiHand", "training", args, device=dev)
flat = FlatParams(model); opt = FusedAdam(flat, lr=1e-6)
bufs = [b.clone() for b in model.buffers()]
terms = list(args.losses)
names = [n for n, _ in model.named_parameters()]
assert len(names) == len(flat.params)
offs = []
o = 0
for p in flat.params:
    offs.append((o, o + p.numel())); o += p.numel()


def restore():
    with torch.no_grad():
        for b, s in zip(model.buffers(), bufs):
            b.copy_(s)


def run_eager():
    restore()
    loss, dic = forward_backward(model, LossFunction(), opt, ex, args)
    torch.cuda.synchronize()
    return {k: dic[k].detach().clone() for k in terms}, flat.grad.clone()


def compare(tag, a, b):
    (da, ga), (db, gb) = a, b
    bad_terms = [k for k in terms if not torch.equal(da[k], db[k])]
    bad = []
    for n, (lo, hi) in zip(names, offs):
        x, y = ga[lo:hi], gb[lo:hi]
        if not torch.equal(x, y):
            bad.append((float((x - y).abs().max() / x.abs().max().clamp_min(1e-30)), n))
    print(f"{tag}: loss terms differing bitwise: {bad_terms or 'none'}; parameters with differing gradients: {len(bad)} of {len(names)}")
    for r, n in sorted(bad, reverse=True)[:400]:
        print(f"    {r:.2e}  {n}")


run_eager()          # warm-up: the FIRST eager step of a model runs before the weight re-layout has its Winograd-domain filters (other kernels)
eager = [run_eager() for _ in range(RUNS)]
for i in range(1, RUNS):
    compare(f"eager 0 vs eager {i}", eager[0], eager[i])

_prime_for_capture(dev)
static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in ex.items()}
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        forward_backward(model, LossFunction(), opt, static, args)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
restore()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    gl, gdic = forward_backward(model, LossFunction(), opt, static, args)
graph = []
for _ in range(RUNS):
    restore(); g.replay(); torch.cuda.synchronize()
    graph.append(({k: gdic[k].detach().clone() for k in terms}, flat.grad.clone()))
for i in range(1, RUNS):
    compare(f"graph 0 vs graph {i}", graph[0], graph[i])
compare("eager 0 vs graph 0", eager[0], graph[0])
print("loss terms eager:", {k: float(v) for k, v in eager[0][0].items()})
print("loss terms graph:", {k: float(v) for k, v in graph[0][0].items()})
