"""Flat parameter / gradient buffers and the fused Adam step.

MI355X-first layout: all trainable parameters of the model live in ONE contiguous fp32 buffer and all their
gradients in another (parameters and .grad are views), so that
  * zero_grad is one memset, the optimizer step is one HBM-bound kernel launch (hifihr_adam_step), and
  * the data-parallel all-reduce runs on contiguous slices of the gradient buffer with no packing copies
    (hifihr_amd/dist.py).
Semantics: torch.optim.Adam as configured at reference train_hrnet.py:546-551.
"""
from __future__ import annotations

import os

import torch

from ._lib import get_lib, require_cuda


class FlatParams:
    """Re-homes module parameters into a flat buffer (registration order) and pins their .grad to views of a
    flat gradient buffer."""

    def __init__(self, module: torch.nn.Module, align: int = 64):
        self.params = [p for p in module.parameters() if p.requires_grad]
        dev = self.params[0].device
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + align - 1) // align * align        # keep every tensor 256-byte aligned
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, self.offsets):
            p.data = self._view(self.flat, p, o, init=p.data)
            p.grad = self._view(self.grad, p, o)
            p._hifihr_direct_grad = True          # kernels may accumulate straight into this (pre-zeroed) buffer

    @staticmethod
    def _view(buf, p, o, init=None):
        """A view of buf[o:o+n] with p's logical shape; conv weights kept in channels_last (physical [K][R][S][C]) stay so."""
        n = p.numel()
        seg = buf[o:o + n]
        if p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last) and not p.is_contiguous():
            K, C, R, S = p.shape
            v = seg.view(K, R, S, C).permute(0, 3, 1, 2)
        else:
            v = seg.view(p.shape)
        if init is not None:
            v.copy_(init)
        return v

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):                # re-pin (a None grad would detach the view)
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self._view(self.grad, p, o)

    def param_count(self):
        return sum(p.numel() for p in self.params)


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics on a FlatParams (one kernel launch per step).  Subclasses Optimizer so that
    torch.optim.lr_scheduler.MultiStepLR (train_hrnet.py:551) drives param_groups[0]['lr'] unchanged."""

    def __init__(self, flat: FlatParams, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
        self.flatp = flat
        super().__init__([flat.flat], dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.exp_avg = torch.zeros_like(flat.flat)
        self.exp_avg_sq = torch.zeros_like(flat.flat)
        self._step_count = 0
        self.grad_scale = grad_scale
        self._lib = None
        # graph mode: the step counter and the learning rate live in device memory (hifihr_adam_step_counted: the kernel derives the bias
        # corrections and advances the counter itself), uploaded by prepare_step() only when the host's view of them changed -- a scheduler
        # step, a restored checkpoint.  HIFIHR_ADAM_COUNTED=0: the two per-step scalars uploaded before every replay (hifihr_adam_step_dyn).
        self.graph_mode = False
        self._counted = os.environ.get("HIFIHR_ADAM_COUNTED", "1") != "0"
        self._state = None                  # device image of (lr, betas, step)
        self._state_sig = None              # (lr, betas, completed steps) the device holds
        self._dyn = None
        self._dyn_host = None
        self._prepared = False              # prepare_step() ran and the launch / replay it announced has not been noted yet

    @property
    def step_count(self):
        return self._step_count

    @step_count.setter
    def step_count(self, value):
        # (a restored snapshot / checkpoint: the device's counter is refreshed by the next prepare_step)
        self._step_count = int(value)
        self._state_sig = None
        self._prepared = False

    _RING = 32        # pinned staging slots for the per-step scalars

    def enable_graph_mode(self):
        """The device-side state is allocated ONCE per optimizer: a graph captured after an earlier enable_graph_mode() has the
        buffers' addresses baked in, and a second call (a re-capture after a lambda schedule fired) must not free them under it."""
        self.graph_mode = True
        self._prepared = False
        self._state_sig = None
        if self._counted and self.flatp.flat.is_cuda:
            if self._lib is None:
                self._lib = get_lib()
            if self._state is None:
                self._state = torch.zeros(int(self._lib.c.hifihr_adam_state_bytes()), dtype=torch.uint8, device=self.flatp.flat.device)
        else:
            self._counted = False
        if self._dyn is None:
            self._dyn = torch.zeros(2, device=self.flatp.flat.device)
            cuda = self.flatp.flat.is_cuda
            self._dyn_host = torch.zeros(self._RING, 2).pin_memory() if cuda else torch.zeros(self._RING, 2)
            self._dyn_events = [None] * self._RING

    def disable_graph_mode(self):
        """Back to the eager step (scalars passed by value, step counter advanced by step())."""
        self.graph_mode = False
        self._prepared = False
        self._state_sig = None              # whatever the device counter holds is re-uploaded if graph mode comes back

    def note_step_done(self):
        """The launch / replay announced by the last prepare_step() has been enqueued (GraphedTrainStep.__call__ after graph.replay(),
        step() for an eager launch in graph mode).  A prepare_step() that finds the previous one un-noted -- the replay never ran:
        an exception, a caller that prepared twice -- takes its count back and re-uploads the device state, so the host counter
        (bias correction, state_dict()['step']) stays the number of steps that were actually enqueued."""
        self._prepared = False

    def prepare_step(self):
        """Graph mode: advance the step counter and upload {lr/(1-b1^t), 1/sqrt(1-b2^t)}; call before each replay.
        The asynchronous copy reads a pinned host slot when the GPU gets to it, possibly many host steps later: every step
        writes a slot of its own (ring), and a slot is rewritten only after the copy that read it has completed (event) --
        a single staging buffer let a pending copy pick up a LATER step's scalars when the host ran ahead."""
        g = self.param_groups[0]
        if self._prepared:                  # the step announced last time was never enqueued
            self._step_count -= 1
            self._state_sig = None
        self._prepared = True
        self._step_count += 1
        b1, b2 = g["betas"]
        if self._counted:
            # the device advances its own counter: upload only when what it holds is not what this step needs
            want = (float(g["lr"]), float(b1), float(b2), self._step_count - 1)
            if self._state_sig != want:
                self._state.copy_(self._lib.adam_state_image(*want))        # (pageable source: a blocking, stream-ordered copy; rare)
            self._state_sig = (want[0], want[1], want[2], self._step_count)     # after the replay that follows
            return
        slot = self.step_count % self._RING
        ev = self._dyn_events[slot]
        if ev is not None:
            ev.synchronize()
        self._dyn_host[slot, 0] = g["lr"] / (1.0 - b1 ** self.step_count)
        self._dyn_host[slot, 1] = 1.0 / (1.0 - b2 ** self.step_count) ** 0.5
        self._dyn.copy_(self._dyn_host[slot], non_blocking=True)
        if self._dyn.is_cuda:
            ev = self._dyn_events[slot] or torch.cuda.Event()
            ev.record()
            self._dyn_events[slot] = ev

    def zero_grad(self, set_to_none: bool = False):
        self.flatp.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        if self._lib is None:
            self._lib = get_lib()
        require_cuda(self.flatp.flat)
        g = self.param_groups[0]
        if self.graph_mode:
            capturing = self.flatp.flat.is_cuda and torch.cuda.is_current_stream_capturing()
            if not capturing:               # an eager launch in graph mode (the warm-up steps): it consumes its prepare_step()
                if not self._prepared:
                    raise RuntimeError("FusedAdam.step() in graph mode without prepare_step(): the device-side step counter / the uploaded "
                                       "bias corrections would be those of the previous step")
                self._prepared = False
        if self.graph_mode and self._counted:
            self._lib.adam_step_counted(self.flatp.flat, self.flatp.grad, self.exp_avg, self.exp_avg_sq, self.grad_scale, g["eps"],
                                        g["weight_decay"], self._state)
            return
        if self.graph_mode:
            self._lib.adam_step_dyn(self.flatp.flat, self.flatp.grad, self.exp_avg, self.exp_avg_sq, self.grad_scale,
                                    g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self._dyn)
            return
        self._step_count += 1
        from .ops import PROFILE
        PROFILE.bracket("adam", lambda: self._lib.adam_step(self.flatp.flat, self.flatp.grad, self.exp_avg, self.exp_avg_sq,
                                                            self.grad_scale, g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                                                            g["weight_decay"], self.step_count))

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)
