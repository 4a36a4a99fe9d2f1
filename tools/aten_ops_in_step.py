"""Which ATen operators still launch kernels inside the training step: torch.profiler over 3 eager steps of the headline configuration,
operators with device time listed with their call counts per step and the Python frames that called them.
usage: python tools/aten_ops_in_step.py"""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from hifihr_amd import options, synth  # noqa: E402
from hifihr_amd.losses import LossFunction  # noqa: E402
from hifihr_amd.mano_tables import synthetic_mano_tables  # noqa: E402
from hifihr_amd.models import Model  # noqa: E402
from hifihr_amd.optim import FlatParams, FusedAdam  # noqa: E402
from hifihr_amd.traineval import data_dic, train_step  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream())
args = options.make_args()
args.losses = ["joint_3d", "vert_3d", "mpose", "mshape", "edge_length", "sil", "texture", "mrgb", "ssim_tex"]
mt = synthetic_mano_tables(0)
model = Model(True, dev, False, "mano", False, "res18", mano_tables=mt).to(dev).train()
flat = FlatParams(model)
opt = FusedAdam(flat, lr=1e-4)
lf = LossFunction()
from hifihr_amd import ops  # noqa: E402
batch = synth.make_batch(model.hand_layer.handle, model.renderer_p3d, 32, device=dev)
ex = data_dic(batch, "FreiHand", "training", args, device=dev)
for _ in range(3):
    train_step(model, lf, opt, ex, args)
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(N):
        train_step(model, lf, opt, ex, args)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_stack_n=6):
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = e.self_cuda_time_total
    if dt <= 0 or not e.key.startswith("aten::"):
        continue
    stack = [s for s in e.stack if "/root/repo" in s or "hifihr_amd" in s or "train" in s][:3]
    rows.append((dt / N, e.count / N, e.key, stack))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"ATen operators with kernels of their own: {tot:.0f} us of device time per step, {sum(r[1] for r in rows):.0f} calls per step")
for dt, cnt, key, stack in rows:
    where = " <- ".join(s.split("/")[-1] for s in stack) if stack else "(autograd engine / no Python frame)"
    print(f"{dt:8.1f} us  x{cnt:4.1f}  {key:28s} {where}")
