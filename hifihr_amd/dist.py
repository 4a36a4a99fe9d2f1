"""Data-parallel gradient exchange: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The reference's only parallel construct is nn.DataParallel (reference train_hrnet.py:560).
Here each rank holds a full replica and owns samples [rank*B, (rank+1)*B) of every global batch; the single
exchange per step is a SUM all-reduce of the flat fp32 gradient buffer (hifihr_amd/optim.FlatParams), cut
into a few contiguous buckets that are launched asynchronously as soon as backward has produced every
gradient of a bucket (reverse registration order ~ backward order), so the ring all-reduce overlaps the rest
of backward.  The 1/world factor is folded into the fused Adam step (FusedAdam.grad_scale).  xGMI is
point-to-point (7 links x ~153 GB/s per GPU): ~50 MB of gradients per step in 4 buckets keeps each message
large enough to run at link bandwidth.  Works unchanged on the gloo backend (CPU tests).
"""
from __future__ import annotations

import datetime
import os

import torch
import torch.distributed as dist

from .optim import FlatParams


def init_process_group_from_env(backend: str | None = None, timeout_min: float | None = None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).  Returns (rank, local_rank, world).
    `timeout_min` (or HIFIHR_DIST_TIMEOUT_MIN): the process group's collective timeout.  Every collective runs under it, barriers
    included, so it must cover the longest stretch one rank works alone (rank 0's evaluation pass between epochs)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # HIFIHR_DIST_BACKEND=gloo: functional runs of the multi-process path on fewer GPUs than ranks (RCCL refuses two ranks
            # on one device); never used for measurements
            backend = os.environ.get("HIFIHR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        if timeout_min is None and os.environ.get("HIFIHR_DIST_TIMEOUT_MIN"):
            timeout_min = float(os.environ["HIFIHR_DIST_TIMEOUT_MIN"])
        kw = {} if timeout_min is None else {"timeout": datetime.timedelta(minutes=float(timeout_min))}
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


class GradReducer:
    def __init__(self, flat: FlatParams, num_buckets: int = 4, group=None, first_fraction: float = 0.08):
        """Buckets are contiguous in registration (= forward) order, so backward completes them last-to-first and the exchange of
        bucket 0 is the only one that cannot hide behind backward work: it gets `first_fraction` of the elements (the stem and the
        first stages are small anyway), the other buckets share the rest equally."""
        self.flat, self.group = flat, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        n = len(flat.params)
        total = flat.param_count()
        if num_buckets > 1:
            fr = [first_fraction] + [(1.0 - first_fraction) / (num_buckets - 1)] * (num_buckets - 1)
        else:
            fr = [1.0]
        cuts, c = [], 0.0
        for f in fr[:-1]:
            c += f
            cuts.append(c * total)                     # cumulative element targets of the bucket ends
        bounds, acc = [0], 0
        for i, p in enumerate(flat.params):
            acc += p.numel()
            if len(bounds) - 1 < len(cuts) and acc >= cuts[len(bounds) - 1] and i + 1 < n:
                bounds.append(i + 1)
        bounds.append(n)
        self.buckets = []                                          # (param lo, param hi, elem lo, elem hi)
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            e_lo = flat.offsets[lo]
            e_hi = flat.offsets[hi] if hi < n else flat.numel
            self.buckets.append((lo, hi, e_lo, e_hi))
        self.param_bucket = {}
        for b, (lo, hi, _, _) in enumerate(self.buckets):
            for i in range(lo, hi):
                self.param_bucket[i] = b
        self._pending = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._done = [False] * n                         # a parameter may be reported twice (autograd hook + direct writer)
        self._handles = []
        self._paused = False
        self._trace = [] if os.environ.get("HIFIHR_DP_TRACE") else None       # debugging: (event, parameter index, bucket, pending)
        if self.world > 1:
            for i, p in enumerate(flat.params):
                hook = self._make_hook(i)
                p.register_post_accumulate_grad_hook(hook)
                # kernels that accumulate straight into the flat gradient buffer (conv wgrad, BN dgamma/dbeta) bypass
                # autograd's AccumulateGrad, so hifihr_amd.ops calls this instead once the parameter's gradient is enqueued
                p._hifihr_grad_ready = hook
        self.reset()

    def reset(self):
        for b, (lo, hi, _, _) in enumerate(self.buckets):
            self._pending[b] = hi - lo
            self._launched[b] = False
        self._done = [False] * len(self._done)
        self._handles = []

    def _launch(self, b):
        _, _, e_lo, e_hi = self.buckets[b]
        self._launched[b] = True
        if self.flat.grad.is_cuda:
            # part of the bucket's gradients may have been enqueued on a side-branch stream (ops.side_branch: the light estimator runs beside
            # the main chain); the collective is ordered behind the CURRENT stream only, so make that stream wait for the branches first
            # (and a hook that fires on a BRANCH stream must wait for the step's own stream likewise)
            from .ops import side_branch
            cur = torch.cuda.current_stream(self.flat.grad.device)
            for key, st in side_branch._streams.items():
                if key[0] != self.flat.grad.device:
                    continue
                for other in (st, side_branch._home.get(key)):
                    if other is not None and other.cuda_stream != cur.cuda_stream:
                        cur.wait_stream(other)
        self._handles.append(dist.all_reduce(self.flat.grad[e_lo:e_hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _make_hook(self, i):
        def hook(_param):
            if self._paused or self._done[i]:
                return
            if getattr(_param, "_hifihr_grad_deferred", False):
                # autograd fires this hook when the function that took the parameter as an input has run its backward -- also when it
                # returned None for it.  A fused function that handed the batch-norm apply (and with it dgamma / dbeta) over to the
                # PRODUCER's backward (ops._WinoLink) has not written the gradient yet: the producer reports it (ops._grad_ready).
                return
            self._done[i] = True
            b = self.param_bucket[i]
            self._pending[b] -= 1
            if self._trace is not None:
                import traceback
                self._trace.append(("ready", i, b, self._pending[b], [f"{f.name}:{f.lineno}" for f in traceback.extract_stack()[-5:-1]]))
            if self._pending[b] == 0 and not self._launched[b]:
                if self._trace is not None:
                    self._trace.append(("launch", b))
                self._launch(b)
        return hook

    def finish(self):
        """Call after loss.backward(): launches buckets whose parameters got no gradient this step, waits for all."""
        if self.world > 1:
            for b in reversed(range(len(self.buckets))):
                if not self._launched[b]:
                    self._launch(b)
            for h in self._handles:
                h.wait()
        self.reset()

    def pause_hooks(self, paused: bool = True):
        """Graph-replayed steps (traineval.GraphedTrainStep) exchange the gradients after the graph: the per-parameter
        hooks must then neither count nor launch anything while backward is being captured."""
        self._paused = paused
        self.reset()

    def all_reduce_params_async(self, lo: int, hi: int):
        """SUM all-reduce of the gradients of parameters [lo, hi) (registration order: one contiguous slice of the flat buffer),
        asynchronous: the collective is ordered behind everything enqueued on the current stream so far and runs beside what is
        enqueued after it (traineval.SegmentedGraphedTrainStep launches the next backward segment meanwhile).  -> handle or None."""
        if self.world <= 1 or hi <= lo:
            return None
        n = len(self.flat.params)
        e_lo = self.flat.offsets[lo]
        e_hi = self.flat.offsets[hi] if hi < n else self.flat.numel
        return dist.all_reduce(self.flat.grad[e_lo:e_hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def all_reduce_flat(self):
        """SUM all-reduce of the whole flat gradient buffer, one asynchronous call per bucket (pipelined), then wait."""
        if self.world > 1:
            handles = [dist.all_reduce(self.flat.grad[e_lo:e_hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                       for (_, _, e_lo, e_hi) in self.buckets]
            for h in handles:
                h.wait()

    def time_buckets(self, reps: int = 5):
        """Average microseconds of one SUM all-reduce per bucket (device events around `reps` back-to-back collectives on a scratch
        copy of the bucket; every rank must call it).  bench.py records it next to the step time when N > 1."""
        if self.world <= 1:
            return []
        out = []
        for (_, _, e_lo, e_hi) in self.buckets:
            buf = torch.zeros(e_hi - e_lo, device=self.flat.grad.device)
            dist.all_reduce(buf, group=self.group)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                dist.all_reduce(buf, group=self.group)
            e1.record()
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) * 1e3 / reps)
        return out

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world


def broadcast_params(flat: FlatParams, src: int = 0, group=None):
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat.flat, src=src, group=group)
