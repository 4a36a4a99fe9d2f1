#!/usr/bin/env python3
"""Eager-vs-eager and eager-vs-graph loss after 3 tiny-lr steps: separates atomics noise from a capture bug."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import test_gpu_e2e as t
from hifihr_amd.losses import LossFunction
from hifihr_amd.models import Model
from hifihr_amd.optim import FlatParams, FusedAdam
from hifihr_amd.traineval import GraphedTrainStep, train_step

torch.cuda.set_stream(torch.cuda.Stream())
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
tables, args, model, ref, ex, ex_cpu = t._setup(B)
sd = {k: v.clone() for k, v in model.state_dict().items()}
def fresh():
    m = Model(True, torch.device("cuda"), False, "mano", False, "res18", mano_tables=tables).cuda().train()
    m.load_state_dict(sd)
    return m
res = []
for trial in range(3):
    m = fresh(); flat = FlatParams(m); opt = FusedAdam(flat, lr=1e-6)
    ls = []
    for _ in range(4):
        l, _ = train_step(m, LossFunction(), opt, ex, args)
        ls.append(float(l))
    res.append(ls); print("eager", trial, ls)
m2 = fresh(); flat2 = FlatParams(m2); opt2 = FusedAdam(flat2, lr=1e-6)
g = GraphedTrainStep(m2, LossFunction(), opt2, ex, args, warmup=3)
lg, _ = g(); torch.cuda.synchronize()
print("graph step4", float(lg))
