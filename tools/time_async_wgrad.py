"""Eager training step with weight gradients on a side stream (ops._AsyncWgrad) vs on the main stream, alternating in one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hifihr_amd import ops, options, synth
from hifihr_amd.losses import LossFunction
from hifihr_amd.mano_tables import synthetic_mano_tables
from hifihr_amd.models import Model
from hifihr_amd.optim import FlatParams, FusedAdam
from hifihr_amd.traineval import data_dic, train_step

dev = torch.device("cuda")
torch.cuda.set_stream(torch.cuda.Stream())
args = options.baseline_config2_args(train_batch=32)
torch.manual_seed(0)
model = Model(True, dev, False, "mano", False, "res18", mano_tables=synthetic_mano_tables(0)).to(dev).train()
flat = FlatParams(model); opt = FusedAdam(flat, lr=1e-6); lf = LossFunction()
ex = data_dic(synth.make_batch(model.hand_layer.handle, model.renderer_p3d, 32, device=dev), "FreiHand", "training", args, device=dev)
for _ in range(5):
    train_step(model, lf, opt, ex, args)
for rnd in range(3):
    for mode in ("1", "0"):
        os.environ["HIFIHR_ASYNC_WGRAD"] = mode
        for _ in range(3):
            train_step(model, lf, opt, ex, args)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30):
            train_step(model, lf, opt, ex, args)
        torch.cuda.synchronize()
        print(f"round {rnd} async={mode}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms/step")
