"""ATen operators that launch kernels inside one eager training step, each with the chain of profiler events it ran under (the autograd
node for backward operators, the Python-side operator for forward ones) -- where tools/aten_ops_in_step.py has no Python frames to show.
usage: python tools/aten_ops_parents.py"""
import os, sys, collections
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
opt = FusedAdam(FlatParams(model), lr=1e-4)
lf = LossFunction()
ex = data_dic(synth.make_batch(model.hand_layer.handle, model.renderer_p3d, 32, device=dev), "FreiHand", "training", args, device=dev)
for _ in range(3):
    train_step(model, lf, opt, ex, args)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    train_step(model, lf, opt, ex, args)
    torch.cuda.synchronize()
rows = collections.Counter()
for e in prof.events():
    if not e.name.startswith("aten::"):
        continue
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = getattr(e, "self_cuda_time_total", 0)
    if dt <= 0:
        continue
    chain, p = [], e.cpu_parent
    while p is not None and len(chain) < 4:
        chain.append(p.name[:60])
        p = p.cpu_parent
    rows[(e.name, " <- ".join(chain) or "(top level)", tuple(e.input_shapes[0]) if e.input_shapes else ())] += 1
for (name, chain, shp), n in sorted(rows.items(), key=lambda kv: -kv[1]):
    print(f"x{n:2d} {name:24s} {chain}")
