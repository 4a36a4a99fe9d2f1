#!/usr/bin/env python3
"""Noise floor of the training step between replicas that start from identical weights and see the identical batch:
R eager replicas + one graphed replica, K steps, every loss term per step.  The spread between EAGER replicas is what float-atomic
ordering alone produces; the graph-vs-eager tests of tests/test_gpu_e2e.py have to sit well above it.
usage: noise_floor.py [B] [images: noise|render|graded] [replicas] [steps] [lr]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from hifihr_amd import options, synth
from hifihr_amd.losses import LossFunction
from hifihr_amd.mano_tables import synthetic_mano_tables
from hifihr_amd.models import Model
from hifihr_amd.optim import FlatParams, FusedAdam
from hifihr_amd.traineval import GraphedTrainStep, data_dic, train_step

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
images = sys.argv[2] if len(sys.argv) > 2 else "noise"
NR = int(sys.argv[3]) if len(sys.argv) > 3 else 4
K = int(sys.argv[4]) if len(sys.argv) > 4 else 4
lr = float(sys.argv[5]) if len(sys.argv) > 5 else 1e-6

torch.cuda.set_stream(torch.cuda.Stream())
dev = torch.device("cuda")
tables = synthetic_mano_tables(0)
args = options.baseline_config2_args(train_batch=B)
torch.manual_seed(0)
model0 = Model(True, dev, False, "mano", False, "res18", mano_tables=tables).to(dev).train()
sd = {k: v.clone() for k, v in model0.state_dict().items()}
sample = synth.make_batch(model0.hand_layer.handle, model0.renderer_p3d, B, first_index=0, device=dev,
                          images="noise" if images == "graded" else images)
if images == "graded":
    from test_gpu_e2e import graded_images
    sample["trans_images"] = graded_images(sample["trans_images"])
ex = data_dic(sample, "FreiHand", "training", args, device=dev)


def fresh():
    m = Model(True, dev, False, "mano", False, "res18", mano_tables=tables).to(dev).train()
    m.load_state_dict(sd)
    return m


terms = list(args.losses) + ["loss"]
runs = []
flats = []
for r in range(NR):
    m = fresh(); flat = FlatParams(m); opt = FusedAdam(flat, lr=lr)
    rows = []
    for _ in range(K):
        l, dic = train_step(m, LossFunction(), opt, ex, args)
        torch.cuda.synchronize()
        rows.append({k: float(dic[k]) for k in terms})
    runs.append(rows); flats.append(flat.flat.clone())
m2 = fresh(); flat2 = FlatParams(m2); opt2 = FusedAdam(flat2, lr=lr)
g = GraphedTrainStep(m2, LossFunction(), opt2, ex, args, warmup=3)
grow = []
for _ in range(K):
    l, dic = g(); torch.cuda.synchronize()
    grow.append({k: float(dic[k]) for k in terms})

print(f"B={B} images={images} replicas={NR} steps={K} lr={lr}")
for s in range(K):
    for k in terms:
        v = [runs[r][s][k] for r in range(NR)]
        spread = max(v) - min(v)
        gd = max(abs(grow[s][k] - x) for x in v)
        ref = max(1.0, abs(v[0]))
        if k == "loss" or spread / ref > 1e-6 or gd / ref > 1e-6:
            print(f"step {s} {k:12s} value {v[0]:.6f}  eager spread {spread / ref:.2e}  graph-vs-eager {gd / ref:.2e}")
d = [float((flats[0] - f).abs().max()) for f in flats[1:]] + [float((flats[0] - flat2.flat).abs().max())]
dm = [float((flats[0] - f).abs().mean()) for f in flats[1:]] + [float((flats[0] - flat2.flat).abs().mean())]
print("weights vs replica 0: max", ["%.2e" % x for x in d], "mean", ["%.2e" % x for x in dm], "(last = graph)")
