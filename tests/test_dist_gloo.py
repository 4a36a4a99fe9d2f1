"""world_size-2 gloo test of the data-parallel path (FlatParams + GradReducer): bucketed async all-reduce of
the flat gradient buffer, launched from post-accumulate hooks, equals the full-batch gradient."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class _DirectGradScale(torch.autograd.Function):
    """Mimics the HIP ops that write a parameter gradient straight into the flat buffer (conv wgrad, BN): y = x * w,
    dw accumulated in place, autograd gets None, the reducer is told through p._hifihr_grad_ready."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        ctx.w_param = w
        return x * w

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        p = ctx.w_param
        assert getattr(p, "_hifihr_direct_grad", False) and p.grad is not None
        p.grad.add_((gy * x).sum(0))
        cb = getattr(p, "_hifihr_grad_ready", None)
        if cb is not None:
            cb(p)
        return gy * w, None


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.net = torch.nn.Sequential(torch.nn.Linear(20, 64), torch.nn.ReLU(), torch.nn.Linear(64, 33), torch.nn.ReLU(),
                                       torch.nn.Linear(33, 5))
        self.scale = torch.nn.Parameter(torch.ones(5) * 1.5)          # gradient written directly, not by autograd
        self.unused = torch.nn.Linear(7, 3)                            # a head that gets no gradient (like trans_reg)

    def forward(self, x):
        return _DirectGradScale.apply(self.net(x), self.scale)


def _model():
    torch.manual_seed(3)
    return _Net()


def _model_old():
    torch.manual_seed(3)
    m = torch.nn.Sequential(torch.nn.Linear(20, 64), torch.nn.ReLU(), torch.nn.Linear(64, 33), torch.nn.ReLU(),
                            torch.nn.Linear(33, 5))
    unused = torch.nn.Linear(7, 3)                       # a head that gets no gradient (like trans_reg / scale_reg)
    return torch.nn.ModuleDict({"net": m, "unused": unused})


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from hifihr_amd import dist as hdist
    from hifihr_amd.optim import FlatParams
    hdist.init_process_group_from_env("gloo")
    model = _model()
    flat = FlatParams(model)
    if rank == 1:
        flat.flat.add_(1.0)                              # diverge, then broadcast must repair
    hdist.broadcast_params(flat)
    red = hdist.GradReducer(flat, num_buckets=3)
    assert len(red.buckets) >= 2 and red.grad_scale == 0.5
    g = torch.Generator().manual_seed(11)
    x = torch.randn(8, 20, generator=g); y = torch.randn(8, 5, generator=g)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    for _ in range(2):                                   # two steps: hooks / buckets must re-arm
        flat.zero_grad()
        loss = torch.nn.functional.mse_loss(model(xs), ys)
        loss.backward()
        red.finish()
    hooked = (flat.grad * red.grad_scale).clone()
    # the graph-replay form of the step (traineval.GraphedTrainStep with a reducer): hooks paused during backward, the whole
    # flat buffer exchanged afterwards -- must give the same gradient
    red.pause_hooks(True)
    flat.zero_grad()
    torch.nn.functional.mse_loss(model(xs), ys).backward()
    red.all_reduce_flat()
    assert torch.allclose(flat.grad * red.grad_scale, hooked, atol=1e-7), "paused hooks + all_reduce_flat differs"
    # the segmented form (traineval.SegmentedGraphedTrainStep): the autograd graph cut at a boundary, backward in two calls, the
    # parameters each call completed exchanged by range while the next call runs -- the same gradient again
    flat.zero_grad()
    h1 = model.net[2](model.net[1](model.net[0](xs)))
    leaf = h1.detach().requires_grad_(True)
    out = _DirectGradScale.apply(model.net[4](model.net[3](leaf)), model.scale)
    torch.nn.functional.mse_loss(out, ys).backward()                    # stops at the boundary
    idx = {id(p): i for i, p in enumerate(flat.params)}
    cut = idx[id(model.net[4].weight)]                                   # parameters [cut, n) belong to the second segment
    handles = [red.all_reduce_params_async(cut, len(flat.params))]
    h1.backward(leaf.grad)                                              # the first segment, beside the exchange
    handles.append(red.all_reduce_params_async(0, cut))
    for h in handles:
        if h is not None:
            h.wait()
    assert torch.allclose(flat.grad * red.grad_scale, hooked, atol=1e-7), "segmented backward + per-range exchange differs"
    red.pause_hooks(False)
    # a gradient that a LATER function's backward writes (ops._WinoLink): autograd's hook for the parameter must not count while it
    # is flagged deferred; the writer reports it
    p0 = flat.params[0]
    red.reset()
    p0._hifihr_grad_deferred = True
    before = list(red._pending)
    p0._hifihr_grad_ready(p0)
    assert red._pending == before, "a deferred parameter was counted"
    p0._hifihr_grad_deferred = False
    p0._hifihr_grad_ready(p0)
    assert red._pending[red.param_bucket[0]] == before[red.param_bucket[0]] - 1
    red.reset()
    ret[rank] = hooked.numpy(), flat.flat.clone().numpy()
    dist.destroy_process_group()


def test_two_rank_bucketed_allreduce_matches_full_batch():
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    from hifihr_amd.optim import FlatParams
    model = _model()
    flat = FlatParams(model)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(8, 20, generator=g); y = torch.randn(8, 5, generator=g)
    flat.zero_grad()
    torch.nn.functional.mse_loss(model(x), y).backward()
    import numpy as np
    for r in (0, 1):
        grad, params = ret[r]
        np.testing.assert_allclose(params, flat.flat.numpy(), atol=0)          # broadcast restored identical replicas
        np.testing.assert_allclose(grad, flat.grad.numpy(), atol=1e-6)         # mean of per-rank means == full-batch mean
