import sys, os
sys.path.insert(0, "/root/repo")
import torch
from torch.profiler import profile, ProfilerActivity
from hifihr_amd import options, synth
from hifihr_amd.losses import LossFunction
from hifihr_amd.mano_tables import synthetic_mano_tables
from hifihr_amd.models import Model
from hifihr_amd.optim import FlatParams, FusedAdam
from hifihr_amd.traineval import data_dic, train_step
dev = torch.device("cuda")
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
args = options.baseline_config2_args(train_batch=32)
tables = synthetic_mano_tables(0)
torch.manual_seed(0)
model = Model(True, dev, False, "mano", False, "res18", mano_tables=tables).to(dev).train()
ex = data_dic(synth.make_batch(model.hand_layer.handle, model.renderer_p3d, 32, device=dev), "FreiHand", "training", args, device=dev)
flat = FlatParams(model); opt = FusedAdam(flat, lr=1e-4); lf = LossFunction()
for _ in range(3): train_step(model, lf, opt, ex, args)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    train_step(model, lf, opt, ex, args)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key.startswith("aten::") and e.device_time_total > 0:
        rows.append((e.device_time_total, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
for r in rows[:80]: print(f"{r[0]:8.1f} us  x{r[1]:3d}  {r[2]:28s} {r[3]}")
