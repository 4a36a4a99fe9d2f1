"""Functional check of both data-parallel step forms with 2 ranks on ONE GPU (HIFIHR_DIST_BACKEND=gloo): after a few steps the
replicas must hold identical parameters, and the averaged gradient must equal the mean of the per-rank gradients.
  HIFIHR_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/dp_sync_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from hifihr_amd import dist as hdist, options, synth
from hifihr_amd.losses import LossFunction
from hifihr_amd.mano_tables import synthetic_mano_tables
from hifihr_amd.models import Model
from hifihr_amd.optim import FlatParams, FusedAdam
from hifihr_amd.traineval import GraphedTrainStep, SegmentedGraphedTrainStep, data_dic, forward_backward, train_step

rank, local_rank, world = hdist.init_process_group_from_env()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
B = int(os.environ.get("DP_CHECK_BATCH", 32))
args = options.baseline_config2_args(train_batch=B)
torch.manual_seed(0)
model = Model(True, dev, False, "mano", False, "res18", mano_tables=synthetic_mano_tables(0)).to(dev).train()
flat = FlatParams(model)
hdist.broadcast_params(flat)
reducer = hdist.GradReducer(flat, num_buckets=4)
opt = FusedAdam(flat, lr=1e-4, grad_scale=reducer.grad_scale)
lf = LossFunction()
ex = data_dic(synth.make_batch(model.hand_layer.handle, model.renderer_p3d, B, first_index=rank * B, device=dev), "FreiHand", "training", args, device=dev)


def params_in_sync(tag):
    mine = flat.flat.detach().clone()
    other = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(other, mine)
    diff = max(float((o - other[0]).abs().max()) for o in other)
    if rank == 0:
        print(f"{tag}: max parameter difference across ranks = {diff:.3e}")
    assert diff == 0.0, tag


# the reduced gradient of the eager form == sum of the per-rank gradients (computed without hooks)
reducer.pause_hooks(True)
forward_backward(model, lf, opt, ex, args)
first = flat.grad.detach().clone()
forward_backward(model, lf, opt, ex, args)
local = flat.grad.detach().clone()
noise = float((local - first).abs().max()) / max(float(local.abs().max()), 1e-12)
summed = local.clone()
dist.all_reduce(summed)
reducer.pause_hooks(False)
forward_backward(model, lf, opt, ex, args)
reducer.finish()
torch.cuda.synchronize()
err = float((flat.grad - summed).abs().max()) / max(float(summed.abs().max()), 1e-12)
if rank == 0:
    print(f"eager hooks: reduced gradient vs sum of local gradients, relative max error = {err:.3e} "
          f"(run-to-run noise of one rank's own gradient: {noise:.3e} -- float atomics, amplified by train-mode batch-norm)")
if not err < max(20 * noise, 1e-5) and rank == 0:          # which parameters were exchanged before their gradient was complete?
    names = {id(p): n for n, p in model.named_parameters()}
    scale = max(float(summed.abs().max()), 1e-12)
    worst = sorted(((float((flat.grad[o:o + p.numel()] - summed[o:o + p.numel()]).abs().max()) / scale, names.get(id(p), "?"))
                    for p, o in zip(flat.params, flat.offsets)), reverse=True)[:8]
    print("parameters whose reduced gradient differs most:", worst)
    if reducer._trace is not None:
        idx = {id(p): i for i, p in enumerate(flat.params)}
        pn = {idx[id(p)]: n for n, p in model.named_parameters() if id(p) in idx}
        for ev in reducer._trace[-400:]:
            if ev[0] == "launch" or "layer3.0" in pn.get(ev[1], "") or "layer3.1.conv1" in pn.get(ev[1], ""):
                print("   trace", ev, pn.get(ev[1], "") if ev[0] == "ready" else "", reducer.buckets[ev[1]] if ev[0] == "launch" else "")
assert err < max(20 * noise, 1e-5), (err, noise)
# ... and == what ONE process computes for the global batch of 2 B with per-replica batch-norm statistics: rank 0 runs every rank's
# half itself (same weights, no hooks) and averages
reducer.pause_hooks(True)
mean_local = torch.zeros_like(flat.grad)
for r in range(world):
    ex_r = ex if r == rank else data_dic(synth.make_batch(model.hand_layer.handle, model.renderer_p3d, B, first_index=r * B, device=dev),
                                       "FreiHand", "training", args, device=dev)
    forward_backward(model, lf, opt, ex_r, args)
    mean_local += flat.grad / world
reducer.pause_hooks(False)
err1 = float((mean_local - summed / world).abs().max()) / max(float(summed.abs().max()) / world, 1e-12)
if rank == 0:
    print(f"single process on the concatenated batch (per-replica batch-norm) vs the averaged exchanged gradient: {err1:.3e}")
assert err1 < max(20 * noise, 1e-5), (err1, noise)
for _ in range(3):
    train_step(model, lf, opt, ex, args, backward_hook=reducer.finish)
params_in_sync("eager form, 3 steps")
g = GraphedTrainStep(model, lf, opt, ex, args, reducer=reducer)
for _ in range(3):
    g()
params_in_sync("graph form, 3 steps")
# third form (round 3): backward as separate graph launches per trunk segment, each segment's bucket exchanged while the next one runs
g.release()
sg = SegmentedGraphedTrainStep(model, lf, opt, ex, args, reducer)
reducer.pause_hooks(True)
forward_backward(model, lf, opt, ex, args)
whole = flat.grad.detach().clone()
sg._eager_segmented()
seg = flat.grad.detach().clone()
e_seg = float((seg - whole).abs().max()) / max(float(whole.abs().max()), 1e-12)
if rank == 0:
    print(f"segmented backward vs whole backward (local gradient): relative max difference = {e_seg:.3e}")
assert e_seg < max(50 * noise, 5e-5), (e_seg, noise)
for _ in range(3):
    sg()
params_in_sync("segmented graph form, 3 steps")
sg.release()
reducer.pause_hooks(False)
for _ in range(2):
    train_step(model, lf, opt, ex, args, backward_hook=reducer.finish)
params_in_sync("eager form again")
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print("dp sync check ok")
