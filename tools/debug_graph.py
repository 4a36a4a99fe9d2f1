import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from hifihr_amd import options, synth
from hifihr_amd.losses import LossFunction
from hifihr_amd.mano_tables import synthetic_mano_tables
from hifihr_amd.models import Model
from hifihr_amd.optim import FlatParams, FusedAdam
from hifihr_amd.traineval import data_dic, train_step, GraphedTrainStep
dev = torch.device("cuda")
tables = synthetic_mano_tables(0)
args = options.baseline_config2_args(train_batch=8)
torch.manual_seed(0)
model = Model(True, dev, False, "mano", False, "res18", mano_tables=tables).to(dev).train()
flat = FlatParams(model); opt = FusedAdam(flat, lr=1e-6)
ex = data_dic(synth.make_batch(model.hand_layer.handle, model.renderer_p3d, 8, device=dev), "FreiHand", "training", args, device=dev)
for i in range(2):
    l, d = train_step(model, LossFunction(), opt, ex, args)
    print("eager", i, float(l), bool(torch.isfinite(flat.flat).all()))
g = GraphedTrainStep(model, LossFunction(), opt, ex, args, warmup=2)
print("after capture: params finite", bool(torch.isfinite(flat.flat).all()), "dyn", g.opt._dyn.tolist(), "grad finite", bool(torch.isfinite(flat.grad).all()))
for i in range(3):
    l, d = g()
    torch.cuda.synchronize()
    print("replay", i, float(l), {k: float(v) for k, v in d.items() if k != "loss"}, "params finite", bool(torch.isfinite(flat.flat).all()),
          "grad finite", bool(torch.isfinite(flat.grad).all()), "dyn", g.opt._dyn.tolist())
