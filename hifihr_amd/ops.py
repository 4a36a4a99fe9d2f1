"""torch.autograd bindings of the HIP hot-path kernels (C ABI in include/hifihr.h).

Every op here runs ONLY on the GPU through libhifihr.so; a CPU tensor raises (there is no fallback).
"""
from __future__ import annotations

import os
import threading

import torch

from ._lib import get_lib, require_cuda


import os as _os

_DEBUG_SYNC = _os.environ.get("HIFIHR_DEBUG_SYNC", "0") == "1"


class _Profile:
    """Optional HIP-event brackets around the C-ABI launches (recorded on the stream the kernels are launched on,
    i.e. torch's current stream).  No synchronisation until summary() is called."""

    def __init__(self):
        self.on = False
        self.events = []
        self.last_render = None
        self.conv_log = []          # (geometry, "fwd" | "dgrad") of every implicit-GEMM launch while enabled (bench.py roofline)

    def enable(self):
        self.on, self.events = True, []

    def disable(self):
        self.on = False

    def bracket(self, name, fn):
        if _DEBUG_SYNC:                      # HIFIHR_DEBUG_SYNC=1: name every launch and synchronise after it
            print(f"[hifihr] {name} ...", flush=True)
            r = fn()
            torch.cuda.synchronize()
            print(f"[hifihr] {name} done", flush=True)
            return r
        if not self.on:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        self.events.append((name, e0, e1))
        return r

    def summary(self):
        torch.cuda.synchronize()
        acc = {}
        for name, e0, e1 in self.events:
            t, n = acc.get(name, (0.0, 0))
            acc[name] = (t + e0.elapsed_time(e1) * 1e3, n + 1)
        return {k: (t / n, n) for k, (t, n) in acc.items()}


PROFILE = _Profile()


class ManoLayerHandle:
    """Device-resident MANO tables (replaces the buffers ManoLayer.__init__ registers, my_mano.py:283-313)."""

    def __init__(self, tables):
        self.lib = get_lib()
        self.tables = tables
        self.h = self.lib.mano_create(tables)

    def __del__(self):
        try:
            self.lib.mano_destroy(self.h)
        except Exception:
            pass


class _ManoLBS(torch.autograd.Function):
    @staticmethod
    def forward(ctx, handle, pose, beta):
        require_cuda(pose, beta)
        pose = pose.contiguous().float()
        beta = beta.contiguous().float()
        B = pose.shape[0]
        verts = torch.empty(B, 778, 3, device=pose.device)
        jtr = torch.empty(B, 21, 3, device=pose.device)
        saved = torch.empty(B, 778, 3, device=pose.device)
        PROFILE.bracket("mano_lbs_fwd", lambda: handle.lib.mano_lbs_fwd(handle.h, pose, beta, verts, jtr, saved))
        ctx.handle = handle
        ctx.save_for_backward(pose, beta, saved)
        ctx.set_materialize_grads(False)                      # an unused output's gradient arrives as None, not as a zero-filled tensor
        return verts, jtr

    @staticmethod
    def backward(ctx, gverts, gjtr):
        pose, beta, saved = ctx.saved_tensors
        B = pose.shape[0]
        gpose = torch.empty(B, 48, device=pose.device)
        gbeta = torch.empty(B, 10, device=pose.device)
        gv = gverts.contiguous() if gverts is not None else None
        gj = gjtr.contiguous() if gjtr is not None else None
        PROFILE.bracket("mano_lbs_bwd", lambda: ctx.handle.lib.mano_lbs_bwd(ctx.handle.h, pose, beta, saved, gv, gj, gpose, gbeta))
        return None, gpose, gbeta


def mano_lbs(handle: ManoLayerHandle, pose, beta):
    """ManoLayer.forward (reference utils/my_mano.py:315-483): pose [B,48], beta [B,10] -> verts, jtr."""
    return _ManoLBS.apply(handle, pose, beta)


class _ManoJoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, handle, verts, root_id):
        require_cuda(verts)
        verts = verts.contiguous()
        B = verts.shape[0]
        joints_rel = torch.empty(B, 21, 3, device=verts.device)
        verts_rel = torch.empty(B, 778, 3, device=verts.device)
        root = torch.empty(B, 3, device=verts.device)
        PROFILE.bracket("mano_joints_fwd", lambda: handle.lib.mano_joints_fwd(handle.h, verts, root_id, joints_rel, verts_rel, root))
        ctx.handle, ctx.root_id, ctx.B = handle, root_id, B
        ctx.set_materialize_grads(False)
        return joints_rel, verts_rel, root

    @staticmethod
    def backward(ctx, gj, gv, gr):
        if gj is None and gv is None and gr is None:
            return None, None, None
        dev = (gj if gj is not None else gv if gv is not None else gr).device
        gverts = torch.empty(ctx.B, 778, 3, device=dev)
        c = lambda t: t.contiguous() if t is not None else None
        PROFILE.bracket("mano_joints_bwd", lambda: ctx.handle.lib.mano_joints_bwd(ctx.handle.h, c(gj), c(gv), c(gr), ctx.root_id, gverts))
        return None, gverts, None


class _ManoFull(torch.autograd.Function):
    """ManoLayer.forward + xyz_from_vertice + root-relative step + camera-space offset (hifihr_mano_full_fwd / _bwd): two launches
    forward, ONE backward.  The separate forms above cost two launches forward, two backward, an elementwise add for verts_cam and, in
    backward, its AddBackward, the accumulation of verts_rel's two gradients and two zero fills of unused outputs."""

    @staticmethod
    def forward(ctx, handle, pose, beta, root_id, root_xyz):
        require_cuda(pose, beta)
        pose, beta = pose.contiguous().float(), beta.contiguous().float()
        B, dev = pose.shape[0], pose.device
        f = lambda *shape: torch.empty(*shape, device=dev)
        verts, joints_rel, verts_rel, root, saved = f(B, 778, 3), f(B, 21, 3), f(B, 778, 3), f(B, 3), f(B, 778, 3)
        rx = root_xyz.reshape(B, 3).contiguous().float() if root_xyz is not None else None
        verts_cam = f(B, 778, 3)                                # (root_xyz None: = verts_rel)
        PROFILE.bracket("mano_lbs_fwd", lambda: handle.lib.mano_full_fwd(handle.h, pose, beta, root_id, rx, verts, joints_rel, verts_rel,
                                                                         verts_cam, root, saved))
        ctx.handle, ctx.root_id = handle, root_id
        ctx.save_for_backward(pose, beta, saved)
        ctx.set_materialize_grads(False)                        # unused outputs: None, not zero-filled tensors
        # pose / beta are handed back as OUTPUTS (aliases): a consumer that reads them from here -- the mpose / mshape regularisers -- sends
        # its gradient into THIS node's backward, where the kernel adds it to the layer's own (autograd otherwise accumulates the two
        # gradients of each head output with an elementwise launch of its own)
        return joints_rel, verts_rel, verts_cam, root, pose.view_as(pose), beta.view_as(beta)

    @staticmethod
    def backward(ctx, gj, gv, gc, gr, gp_add, gb_add):
        pose, beta, saved = ctx.saved_tensors
        B = pose.shape[0]
        gpose, gbeta = torch.empty(B, 48, device=pose.device), torch.empty(B, 10, device=pose.device)
        c = lambda t: t.contiguous() if t is not None else None
        PROFILE.bracket("mano_lbs_bwd", lambda: ctx.handle.lib.mano_full_bwd(ctx.handle.h, pose, beta, saved, c(gj), c(gv), c(gc), c(gr),
                                                                             ctx.root_id, gpose, gbeta, c(gp_add), c(gb_add)))
        return None, gpose, gbeta, None, None


def mano_full(handle: ManoLayerHandle, pose, beta, root_id=9, root_xyz=None):
    """-> joints_rel [B,21,3], verts_rel [B,778,3], verts_cam [B,778,3] (= verts_rel + root_xyz), pred_root [B,3], and aliases of pose /
    beta for their other consumers (see _ManoFull.forward)."""
    return _ManoFull.apply(handle, pose, beta, root_id, root_xyz)


def mano_joints_root_relative(handle: ManoLayerHandle, verts, root_id=9):
    """xyz_from_vertice(verts).permute(1,0,2) + the root-relative step (models_res_nimble.py:153,160-166).
    -> joints_rel [B,21,3], verts_rel [B,778,3], pred_root [B,3]."""
    return _ManoJoints.apply(handle, verts, root_id)


class LbsHandle:
    """Device-resident tables of a generic skinned mesh (csrc/lbs.hip): v_template [V,3], shapedirs [V,3,S], J_regressor [J,V],
    weights [V,J] (<= 8 non-zeros per vertex), parents [J]."""

    def __init__(self, v_template, shapedirs, j_regressor, weights, parents):
        self.lib = get_lib()
        self.V, self.J, self.S = int(v_template.shape[0]), int(weights.shape[1]), int(shapedirs.shape[2])
        self.h = self.lib.lbs_create(v_template, shapedirs, j_regressor, weights, parents)

    def __del__(self):
        try:
            self.lib.lbs_destroy(self.h)
        except Exception:
            pass


class _Lbs(torch.autograd.Function):
    @staticmethod
    def forward(ctx, handle, theta, beta):
        require_cuda(theta, beta)
        theta, beta = theta.contiguous().float(), beta.contiguous().float()
        B = theta.shape[0]
        assert theta.shape == (B, handle.J, 3) and beta.shape == (B, handle.S)
        verts = torch.empty(B, handle.V, 3, device=theta.device)
        joints = torch.empty(B, handle.J, 3, device=theta.device)
        PROFILE.bracket("lbs_fwd", lambda: handle.lib.lbs_fwd(handle.h, theta, beta, verts, joints))
        ctx.handle = handle
        ctx.save_for_backward(theta, beta)
        return verts, joints

    @staticmethod
    def backward(ctx, gverts, gjoints):
        theta, beta = ctx.saved_tensors
        h, B = ctx.handle, theta.shape[0]
        zero = torch.zeros(B * (h.J * 12 + h.S), device=theta.device)       # one fill: [d(A) scratch | dbeta]
        scratch, gbeta = zero[:B * h.J * 12], zero[B * h.J * 12:].view(B, h.S)
        gtheta = torch.empty(B, h.J, 3, device=theta.device)
        gv = gverts.contiguous() if gverts is not None else torch.zeros(B, h.V, 3, device=theta.device)
        gj = gjoints.contiguous() if gjoints is not None else None
        PROFILE.bracket("lbs_bwd", lambda: h.lib.lbs_bwd(h.h, theta, beta, gv, gj, scratch, gtheta, gbeta))
        return None, gtheta, gbeta


def lbs(handle: LbsHandle, theta, beta):
    """Generic linear-blend skinning: theta [B,J,3], beta [B,S] -> verts [B,V,3], posed joints [B,J,3]."""
    return _Lbs.apply(handle, theta, beta)


class RendererHandle:
    """Replaces MeshRenderer(MeshRasterizer(...), HardPhongShader(...)) of models_res_nimble.py:70-96."""

    def __init__(self, faces, num_verts, image_size=224, aa=3, point_lights=False, **consts):
        self.lib = get_lib()
        self.V, self.H, self.aa = int(num_verts), int(image_size), int(aa)
        self.F = int(len(faces))
        self.h = self.lib.renderer_create(faces, num_verts, image_size=image_size, aa=aa, **consts)
        self.point_lights = bool(point_lights)
        if point_lights:                      # PointLights: `light_dir` of render() is the location (models_res_nimble.py:191-198)
            self.lib.renderer_set_light_mode(self.h, True)

    def set_uv(self, faces_uvs, verts_uvs):
        """TexturesUV tables: faces_uvs [F,3] indices into verts_uvs [n,2] (u, v in [0, 1], v up: PyTorch3D's convention)."""
        self.lib.renderer_set_uv(self.h, faces_uvs, verts_uvs)

    def workspace(self, B, device):
        # one scratch buffer per forward call (it carries the packed vertex records to that call's backward);
        # torch's caching allocator makes this a free-list pop, not a hipMalloc
        return torch.empty(self.lib.render_workspace_bytes(self.h, B), dtype=torch.uint8, device=device)

    def __del__(self):
        try:
            self.lib.renderer_destroy(self.h)
        except Exception:
            pass


class _Render(torch.autograd.Function):
    @staticmethod
    def forward(ctx, handle, verts, vcolors, cam, light_color, light_dir):
        require_cuda(verts, vcolors, cam, light_color, light_dir)
        verts, vcolors, cam = verts.contiguous(), vcolors.contiguous(), cam.contiguous()
        light_color, light_dir = light_color.contiguous(), light_dir.contiguous()
        B, H, S = verts.shape[0], handle.H, handle.H * handle.aa
        rgba = torch.empty(B, 4, H, H, device=verts.device)
        face_id = torch.empty(B, S, S, dtype=torch.int32, device=verts.device)
        ws = handle.workspace(B, verts.device)
        PROFILE.bracket("render_fwd", lambda: handle.lib.render_fwd(handle.h, verts, vcolors, cam, light_color, light_dir, rgba, face_id, ws))
        if PROFILE.on:                       # kept for bench.py's back-to-back timing of the roofline kernel
            PROFILE.last_render = (handle, verts.detach(), vcolors.detach(), cam.detach(), light_color.detach(), light_dir.detach())
        ctx.handle = handle
        ctx.vcol_batched = vcolors.dim() == 3
        ctx.ws = ws
        ctx.save_for_backward(verts, cam, light_color, light_dir, face_id)
        ctx.mark_non_differentiable(face_id)
        ctx.set_materialize_grads(False)
        return rgba, face_id

    @staticmethod
    def backward(ctx, grad_rgba, _):
        if grad_rgba is None:
            return (None,) * 6
        verts, cam, light_color, light_dir, face_id = ctx.saved_tensors
        handle = ctx.handle
        B = verts.shape[0]
        gverts = torch.empty_like(verts)
        need_col = ctx.needs_input_grad[2]
        gvcol = torch.empty_like(verts) if need_col else None
        gl = torch.empty(2, B, 3, device=verts.device)         # adjacent: the library zero-fills both with one launch
        glc, gld = gl[0], gl[1]
        ws = ctx.ws                                  # holds this call's packed vertex records
        g = grad_rgba.contiguous()
        PROFILE.bracket("render_bwd", lambda: handle.lib.render_bwd(handle.h, verts, cam, light_color, light_dir, face_id, g,
                                                                    gverts, gvcol, glc, gld, ws))
        if need_col and not ctx.vcol_batched:
            gvcol = gvcol.sum(0)
        return None, gverts, gvcol, None, glc, gld


class _RenderUV(torch.autograd.Function):
    """The renderer with a TexturesUV texture (RendererHandle.set_uv): maps [B, TH, TW, 3] sampled per sample at the interpolated UV
    (hifihr_render_fwd_uv / _bwd_uv); gradients to the vertices (incl. the path through uv), the texture maps and the light."""

    @staticmethod
    def forward(ctx, handle, verts, maps, cam, light_color, light_dir):
        require_cuda(verts, maps, cam, light_color, light_dir)
        verts, maps, cam = verts.contiguous(), maps.contiguous(), cam.contiguous()
        light_color, light_dir = light_color.contiguous(), light_dir.contiguous()
        B, H, S = verts.shape[0], handle.H, handle.H * handle.aa
        rgba = torch.empty(B, 4, H, H, device=verts.device)
        face_id = torch.empty(B, S, S, dtype=torch.int32, device=verts.device)
        ws = handle.workspace(B, verts.device)
        PROFILE.bracket("render_fwd_uv", lambda: handle.lib.render_fwd_uv(handle.h, verts, maps, cam, light_color, light_dir, rgba, face_id,
                                                                         None, ws))
        ctx.handle, ctx.ws = handle, ws
        ctx.save_for_backward(verts, maps, cam, light_color, light_dir, face_id)
        ctx.mark_non_differentiable(face_id)
        ctx.set_materialize_grads(False)
        return rgba, face_id

    @staticmethod
    def backward(ctx, grad_rgba, _):
        if grad_rgba is None:
            return (None,) * 6
        verts, maps, cam, light_color, light_dir, face_id = ctx.saved_tensors
        handle = ctx.handle
        B = verts.shape[0]
        gverts = torch.empty_like(verts)
        gmaps = torch.zeros_like(maps) if ctx.needs_input_grad[2] else None
        gl = torch.empty(2, B, 3, device=verts.device)
        PROFILE.bracket("render_bwd_uv", lambda: handle.lib.render_bwd_uv(handle.h, verts, maps, cam, light_color, light_dir, face_id,
                                                                         grad_rgba.contiguous(), None, None, gverts, gmaps, gl[0], gl[1],
                                                                         ctx.ws))
        return None, gverts, gmaps, None, gl[0], gl[1]


def render_uv(handle: RendererHandle, verts, maps, cam, light_color, light_dir):
    """renderer_p3d(Meshes(verts, faces, TexturesUV(maps, faces_uvs, verts_uvs)), cameras, lights) + avg_pool2d(aa): the texture image path of
    reference models_res_nimble.py:203-211.  `handle.set_uv(faces_uvs, verts_uvs)` first.  -> rgba [B,4,H,H], face_id."""
    return _RenderUV.apply(handle, verts, maps, cam, light_color, light_dir)


def render(handle: RendererHandle, verts, vcolors, cam, light_color, light_dir):
    """renderer_p3d(meshes, cameras, lights) + avg_pool2d(aa) (models_res_nimble.py:208-211).
    -> rgba [B,4,H,H], face_id int32 [B,H*aa,H*aa]."""
    return _Render.apply(handle, verts, vcolors, cam, light_color, light_dir)


# ------------------------------------------------------------------------------------------------
# convolution on the f32 matrix cores (tensors are logical NCHW in channels_last memory format = physical NHWC)
# ------------------------------------------------------------------------------------------------
_CL = torch.channels_last


class _ZeroPool:
    """Zero-initialised scratch for the self-cleaning batch-norm slot buffers (include/hifihr.h, bn section): `acquire`
    hands out an all-zero tensor; the consumer kernel zeroes it again and `release` returns it for the next producer.
    A buffer that is never consumed (exception between producer and consumer) is simply dropped."""

    def __init__(self):
        self.free = {}

    def acquire(self, n, device):
        lst = self.free.get((n, device))
        if lst:
            return lst.pop()
        return torch.zeros(n, device=device, dtype=torch.float32)

    def release(self, t):
        self.free.setdefault((t.numel(), t.device), []).append(t)


_ZERO_POOL = _ZeroPool()


def _grad_ready(p):
    """Tell the data-parallel reducer (hifihr_amd/dist.py) that this parameter's gradient has been enqueued by a kernel
    that wrote it directly (no autograd AccumulateGrad, hence no post-accumulate hook)."""
    cb = getattr(p, "_hifihr_grad_ready", None)
    if cb is not None:
        cb(p)


class side_branch:
    """`with side_branch(t, "name") as br: y = f(x)` -- f's launches go to a side stream that forks from the current stream here and `br.join(y, ...)`
    (or leaving a graph capture un-joined is an error: call join) joins it again: a chain of small, latency-bound launches that does not depend
    on what the main stream does next overlaps with it -- forward AND backward, because autograd runs a node's backward on the stream of its
    forward; inside a captured step the two become parallel branches of the hipGraph.  Measured (round 5, B = 32): the light estimator
    beside the hand encoder / MANO chain 5.38 -> 5.32 ms/step.  One stream per (device, name), created once.
    `enabled=False` (or HIFIHR_BRANCHES=0) makes the block run inline."""
    _streams = {}
    _home = {}          # (device, name) -> the stream the branch last forked from (the step's own stream)
    _pending = set()    # branches entered since the last join_pending(): their BACKWARD may still be running on the side stream

    def __init__(self, like, name, enabled=True, inputs=()):
        self.on = bool(enabled) and _BRANCHES and like.is_cuda
        self.like, self.name, self.inputs = like, name, inputs

    def __enter__(self):
        if not self.on:
            return self
        dev = self.like.device
        key = (dev, self.name)
        st = side_branch._streams.get(key)
        if st is None:
            st = side_branch._streams[key] = torch.cuda.Stream(device=dev)
            BRANCH_STREAMS.add(st.cuda_stream)
        self.cur, self.side = torch.cuda.current_stream(dev), st
        side_branch._home[key] = self.cur
        side_branch._pending.add(key)
        st.wait_stream(self.cur)
        for t in (self.like,) + tuple(self.inputs):
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(st)
        self._ctx = torch.cuda.stream(st)
        self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            self._ctx.__exit__(*exc)
            if exc[0] is not None:
                self.cur.wait_stream(self.side)
        return False

    def join(self, *outs):
        """The main stream waits for the branch; `outs`: the branch's results that the main stream goes on to read."""
        if self.on:
            for t in outs:
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(self.cur)
            self.cur.wait_stream(self.side)

    @staticmethod
    def join_pending():
        """Join every branch entered since the last call into the current stream (and the stream it forked from).  `join` orders the
        FORWARD; in backward autograd runs the branch's nodes on the side stream again, and the only thing that orders them before what
        follows on the main stream is the first branch node's input gradient reaching a main-stream node.  With a frozen / detached
        trunk there is no such gradient, while the branch's convolutions still write their weight gradients straight into the flat
        gradient buffer (_hifihr_direct_grad: no AccumulateGrad, so autograd does not sync the side stream as a leaf stream either):
        zero_grad / Adam of the main stream would race them, and a capture would end with an un-joined stream.  prepared_weights.__exit__
        calls this after backward, so the optimizer always runs behind the branch."""
        for key in list(side_branch._pending):
            st, home = side_branch._streams.get(key), side_branch._home.get(key)
            if st is not None:
                cur = torch.cuda.current_stream(key[0])
                cur.wait_stream(st)
                if home is not None and home != cur:
                    home.wait_stream(st)
        side_branch._pending.clear()


_BRANCHES = os.environ.get("HIFIHR_BRANCHES", "1") != "0"
_STEM_REDUCE_Y = os.environ.get("HIFIHR_STEM_REDUCE_Y", "1") != "0"       # the stem's batch-norm backward reduction over the pooled grid
_GEMM_PAIR = os.environ.get("HIFIHR_GEMM_PAIR", "1") != "0"
# the same for the 64 -> 64 layers (ResNet layer 1): Winograd F(2x2) data gradient + pixel-reduction weight gradient in one launch
_C64_PAIR = os.environ.get("HIFIHR_C64_PAIR", "1") != "0"
BRANCH_STREAMS = set() # raw handles of side streams that run convolutions BESIDE the main stream (models.Model's light branch)
_CONV_WS = {}          # (device, branch stream or 0) -> zero-initialised, self-cleaning workspace of the balanced convolution schedule
_CONV_WS_BYTES = {}    # (geometry, direction) -> bytes the library wants for it


def _conv_ws(lib, device, geom, bwd):
    """The shared convolution workspace if this shape uses the balanced (stream-K) schedule, else None
    (include/hifihr.h, convolution section).  One buffer per device AND stream serves every layer: launches on one stream are ordered
    and each one hands the buffer back all zero (the light estimator may run on a side stream beside the trunk: models.Model)."""
    key = (geom, bwd)
    nb = _CONV_WS_BYTES.get(key)
    if nb is None:
        nb = lib.conv2d_workspace_bytes(*geom, bwd)
        _CONV_WS_BYTES[key] = nb
    if nb == 0:
        return None
    # (a workspace of its own only for a registered BRANCH stream: keyed by every stream, the capture stream of a graphed step would
    #  get a fresh one allocated -- and zero-filled on every replay -- inside the capture)
    h = torch.cuda.current_stream(device).cuda_stream
    wkey = (device, h if h in BRANCH_STREAMS else 0)
    ws = _CONV_WS.get(wkey)
    if ws is None or ws.numel() * 4 < nb:
        if ws is not None:
            _RETIRED_SCRATCH.append(ws)
        ws = torch.zeros(max(nb, 32 << 20) // 4 + 64, dtype=torch.float32, device=device)
        _CONV_WS[wkey] = ws
    return ws


_WINO_SCRATCH = {}     # (device, name) -> grow-only scratch tensor shared by every Winograd convolution (stream-ordered reuse)


_WGRAD_SLABS = {}       # (device, stream handle) -> per-workgroup slab scratch of the layer-1 / stem weight-gradient kernels


def _wgrad_slabs(device, nbytes):
    """Scratch for hifihr_conv2d_bwd_weight_ws, one buffer per STREAM the launch is issued on (the weight gradients may run on a side
    stream beside the main one, and the captured step on a third): launches on one stream are ordered, so they can share it.  Allocated
    through torch's caching allocator, which is legal inside a stream capture (the block then belongs to the graph's private pool)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    t = _WGRAD_SLABS.get(key)
    if t is None or t.numel() * 4 < nbytes:
        if t is not None:
            _RETIRED_SCRATCH.append(t)
        t = torch.empty((nbytes + 3) // 4, device=device, dtype=torch.float32)
        _WGRAD_SLABS[key] = t
    return t


_RETIRED_SCRATCH = []  # superseded scratch tensors stay allocated: a captured hipGraph may still replay launches that use their addresses


def _wino_scratch(device, name, numel):
    t = _WINO_SCRATCH.get((device, name))
    if t is None or t.numel() < numel:
        if t is not None:
            _RETIRED_SCRATCH.append(t)
        t = torch.empty(numel, device=device, dtype=torch.float32)
        _WINO_SCRATCH[(device, name)] = t
    return t


class _PrecisionStack(threading.local):                # one stack per THREAD: a scope entered in one thread never changes another's dispatch
    def __init__(self):
        self.stack = ["fast"]


_CONV_PRECISION = _PrecisionStack()
_WINOGRAD_ENV = os.environ.get("HIFIHR_WINOGRAD", "1") != "0"          # read once: the process-wide switch (tests set it per subprocess)


class conv_precision:
    """`with conv_precision("reference"):` -- the convolutions called inside run on the DIRECT kernels (implicit GEMM / halo), no Winograd
    transform arithmetic: their outputs then round like a plain fp32 convolution (1e-6 of max |y| instead of Winograd F(4x4, 3x3)'s 1e-5),
    which keeps the ReLU pattern -- and with it the trunk's gradient -- on the reference's side of the discontinuity (README, "Precision of
    the default dispatch").  "fast" (default): Winograd where it pays.  Per call site and per thread, not per process: `Model(conv_precision=...)`
    wraps its encoder in it; each convolution remembers the mode of its forward for its backward.  (HIFIHR_WINOGRAD=0, read at import, is the
    process-wide form.)

    What the knob buys, MEASURED (profiles/r04_precision_by_dispatch.txt, batch-of-8 ResNet-18 fixture): "reference" costs +2.95 ms per
    B = 32 step (5.4 -> 8.3 ms) and brings the FEATURES 2.5 x closer to the reference's (5e-6 of their maximum instead of 1.3e-5); the trunk
    GRADIENTS do not get closer (worst 1.1e-2 against 8.1e-3 on the default dispatch): those differences are ReLU sign flips that any fp32
    summation order other than the reference's own produces.  Use it for feature-level comparisons, not to chase gradient parity."""

    def __init__(self, mode):
        if mode not in ("fast", "reference"):
            raise ValueError(f"conv_precision: 'fast' or 'reference', not {mode!r}")
        self.mode = mode

    def __enter__(self):
        _CONV_PRECISION.stack.append(self.mode)
        return self

    def __exit__(self, *exc):
        _CONV_PRECISION.stack.pop()
        return False


def _wino_allowed():
    return _WINOGRAD_ENV and _CONV_PRECISION.stack[-1] != "reference"


def _wino_ok(C, K, R, S, stride, pad, allowed=None):
    """Winograd F(2x2, 3x3) instead of the direct kernel: stride-1, pad-1 3x3 with >= 128 channels on both sides (measured at
    B = 32, tools/time_wino.py: 295 -> 180 us at 512 channels, 91 -> 69 at 256, 97 -> 81 at 128; HIFIHR_WINOGRAD=0 or
    conv_precision("reference") disable).  `allowed`: the mode a convolution recorded in its forward (its backward must follow it)."""
    if not (_wino_allowed() if allowed is None else allowed):
        return False
    if not (R == 3 and S == 3 and stride == 1 and pad == 1 and C % 32 == 0 and K % 32 == 0):
        return False
    # both sides >= 128 channels -- or (round 4) 64 on one side and >= 128 on the other: VGG19's conv2_1 (64 -> 128 at 112 x 112, three
    # launches of 858 us on the implicit GEMM per config-3 step; HIFIHR_WINO_MIXED=0 keeps it there).  64 -> 64 is conv_wino2_kernel's.
    if C >= 128 and K >= 128:
        return True
    return min(C, K) >= 64 and max(C, K) >= 128 and os.environ.get("HIFIHR_WINO_MIXED", "1") != "0"


def _wino2_fused_ok(lib, N, H, W, C, K, R, S, stride, pad, allowed=None, device=None):
    """The 64 -> 64 stride-1 3x3 layers (ResNet layer 1, VGG19 conv1_2) as register-resident Winograd F(2x2, 3x3), one launch
    (hifihr_conv3x3_c64_wino; HIFIHR_CONV_WINO2=0 keeps them on the direct halo kernel)."""
    return (R == 3 and S == 3 and stride == 1 and pad == 1 and C == 64 and K == 64 and (_wino_allowed() if allowed is None else allowed)
            and lib.conv3x3_c64_wino_supported(N, H, W, C, K) and lib.zero_page_ready(device))   # (no zero page inside a capture: the halo kernel)


_WINO_TILE = {}


def _wino_tile(lib, N, H, W, C, K):
    """Output-tile edge m of the Winograd algorithm this layer runs (hifihr_wino_tile: 4 = F(4x4, 3x3) with 36 positions where the
    batched GEMMs take the shape, else 2 = F(2x2, 3x3) with 16); -> (m, positions, tiles)."""
    key = (N, H, W, C, K)
    e = _WINO_TILE.get(key)
    if e is None:
        m = lib.wino_tile(N, H, W, C, K)
        e = (m, (m + 2) ** 2, lib.wino_tiles(N, H, W, m))        # (tiles: the library's count -- mosaics of 16 images at 14 x 14)
        _WINO_TILE[key] = e
    return e


class _WeightPrep:
    """The per-step re-layouts of convolution weights (transpose for backward-data, Winograd U / U') as ONE launch
    (hifihr_weight_prep) instead of ~40 tiny ones.  The weights change once per optimizer step, so the step brackets its forward
    and backward in `prepared_weights()`: on entry every re-layout registered so far is recomputed into per-layer buffers, inside
    the scope the convolutions pick those up (`get`), outside it (plain op calls, evaluation) they launch their own transform
    as before.  A layer seen for the first time registers itself and is served from the next scope on."""

    def __init__(self):
        self.entries = {}          # (data_ptr, kind) -> [parameter, K, C, RS, kind, buffer]
        self.table, self.njobs, self.dirty, self.active, self.served = None, 0, False, False, set()
        self._retired = []         # superseded job tables stay allocated: a captured hipGraph may still replay a launch that reads one

    def get(self, w, wk, kind):
        """The prepared buffer of `kind` for the parameter `w` (physical [K][R][S][C] = wk), or None (caller does it itself)."""
        if not isinstance(w, torch.nn.Parameter) or wk.data_ptr() != w.data_ptr():
            return None                                        # temporaries (the padded stem weight) have no stable address
        key = (w.data_ptr(), kind)
        e = self.entries.get(key)
        if e is None:
            K, C, R, S = w.shape
            if kind == 5:                                      # channels zero-padded to a multiple of 4 (the 3-channel stem filter)
                n = K * ((C + 3) // 4 * 4) * R * S
            else:
                n = K * C * R * S if kind == 0 else K * C * (16 if kind in (1, 2) else 36)  # kinds 1 / 2: F(2x2) U / U'; 3 / 4: F(4x4)
            self.entries[key] = [w, K, C, R * S, kind, torch.empty(n, device=w.device, dtype=torch.float32)]
            self.dirty = True
            return None
        return e[5] if (self.active and key in self.served) else None

    def begin(self):
        live = {k: e for k, e in self.entries.items() if e[0].data_ptr() == k[0]}      # parameters re-homed since (FlatParams)
        if len(live) != len(self.entries):
            self.entries, self.dirty = live, True
        if not self.entries:
            return
        lib = get_lib()
        if self.dirty:
            jobs = [(e[0], e[5], e[1], e[2], e[3], e[4]) for e in self.entries.values()]
            if self.table is not None:
                self._retired.append((self.table, [e[5] for e in self.entries.values()]))
            self.table, self.njobs = lib.prep_jobs(jobs, jobs[0][0].device), len(jobs)
            self.served, self.dirty = set(self.entries.keys()), False
        PROFILE.bracket("weight_prep", lambda: lib.weight_prep(self.table, self.njobs, 256))
        self.active = True

    def end(self):
        self.active = False


_WEIGHT_PREP = _WeightPrep()


class _DeferredDw:
    """The F(4x4) weight-gradient transforms (dw += G^T dU G, hifihr_wino_dw_transform_parts_m) of a step's layers, collected during
    backward and run as ONE launch when the `prepared_weights()` scope closes -- the weight gradients are read by the optimizer only,
    and ten 5-11 us launches between the backward products (3.9 TB/s each) stream better as one (hifihr_wino4_dw_transform_multi).
    A layer defers only when (a) a scope is open, (b) its gradient is accumulated straight into the flat gradient buffer (nothing is
    returned to autograd), (c) no data-parallel hook waits for the parameter (the bucketed all-reduce overlaps backward: those steps keep
    the immediate launches) and (d) weight gradients are not on the side stream.  A deferred layer's slabs live in a buffer of its own.
    HIFIHR_DEFER_DW=0: immediate launches."""

    def __init__(self):
        self.on = os.environ.get("HIFIHR_DEFER_DW", "1") != "0"
        self.active = False
        self.jobs = []
        self.halo = []
        self.pending = set()               # gradient targets with a job in the lists
        self.early = os.environ.get("HIFIHR_DEFER_EARLY", "1") != "0"
        self.side, self.home, self._streams, self.n_early = None, None, {}, 0

    def wants(self, w, direct):
        # (a layer that already holds a pending job -- a second backward inside one scope -- takes the immediate path: its own slab buffer
        #  must stay as the first backward left it until the flush)
        if self.side is not None and w.grad is not None and w.grad.data_ptr() in self.pending:
            torch.cuda.current_stream(self.side.device).wait_stream(self.side)     # its early flush may still be adding into w.grad beside us
        return (self.on and self.active and direct and not _ASYNC_WGRAD.active and getattr(w, "_hifihr_grad_ready", None) is None
                and (w.grad is None or w.grad.data_ptr() not in self.pending))

    def add(self, dU, parts, tgt, K, C):
        self.jobs.append((dU, int(parts), tgt, int(K), int(C), torch.cuda.current_stream(tgt.device).cuda_stream))
        self.pending.add(tgt.data_ptr())

    def add_halo(self, slabs, nslab, tgt):
        """the slab sum of a 64 -> 64 layer's pixel-reduction weight gradient (hifihr_conv3x3_c64_bwd_pair_slabs left it to us)"""
        self.halo.append((slabs, int(nslab), tgt, torch.cuda.current_stream(tgt.device).cuda_stream))
        self.pending.add(tgt.data_ptr())

    def clear(self):
        self.jobs.clear(); self.halo.clear(); self.pending.clear()
        if self.side is not None:                   # a scope that failed behind its early flush
            if not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream(self.side.device).wait_stream(self.side)
            self.side = None

    def _run(self, jobs, halo):
        lib = get_lib() if (jobs or halo) else None
        if halo:
            PROFILE.bracket("halo_reduce_multi", lambda: lib.conv_halo_wgrad_reduce_multi([h[:3] for h in halo]))
        if jobs:
            PROFILE.bracket("wino_dw_multi", lambda: lib.wino4_dw_transform_multi([j[:5] for j in jobs]))

    def flush(self):
        jobs, self.jobs = self.jobs, []
        halo, self.halo = self.halo, []
        self.pending.clear()
        self._run(jobs, halo)
        if self.side is not None:                   # an early flush is still running beside the step's stream: the optimizer is next
            cur = torch.cuda.current_stream(self.side.device)
            cur.wait_stream(self.side)
            if self.home is not None and self.home != cur:
                self.home.wait_stream(self.side)
            self.side = None

    def flush_early(self, device):
        """Called where the backward has only the stem left (ops._BNReluMaxPool.backward): what the step's layers deferred ON THIS STREAM is
        complete, and nothing but the optimizer reads it -- the one launch goes to a side stream beside the stem's pooling backward and
        weight gradient instead of behind them; `flush` (the scope's exit) joins it.  HIFIHR_DEFER_EARLY=0: only the flush at the exit."""
        if not (self.early and self.active and self.side is None and (self.jobs or self.halo)):
            return
        cur = torch.cuda.current_stream(device)
        h = cur.cuda_stream
        jobs, halo = [j for j in self.jobs if j[5] == h], [j for j in self.halo if j[3] == h]
        if not (jobs or halo):
            return
        self.jobs, self.halo = [j for j in self.jobs if j[5] != h], [j for j in self.halo if j[3] != h]
        st = self._streams.get(device)             # (`pending` keeps these layers until the flush: a second backward of one of them takes the immediate path)
        if st is None:
            st = self._streams[device] = torch.cuda.Stream(device=device)
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            self._run(jobs, halo)
        self.side, self.home, self.n_early = st, cur, self.n_early + 1


_DEFER_DW = _DeferredDw()


class _AsyncWgrad:
    """Weight gradients on a second HIP stream.  In backward the data gradient is the critical path (the previous layer waits for
    it); the weight gradient of a layer is needed only by the optimizer.  Launched on a side stream, the MFMA-bound backward-
    weight kernels run next to the latency-bound kernels of the main stream (batch-norm backward on the small layers, Winograd
    transforms, gradient adds: 10 us kernels that leave the matrix cores idle) instead of queueing between them.  The side
    stream forks after the layer's dy is complete and joins before the optimizer (`join`, called by the step); tensors it
    reads are kept referenced until the join, so neither the caching allocator nor a hipGraph capture needs record_stream.
    Only used while no data-parallel hook can fire in between (the all-reduce of a bucket must not start before its weight
    gradients are complete), and only for gradients accumulated straight into the flat gradient buffer."""

    def __init__(self):
        self.active, self.streams, self.refs = False, {}, []

    def stream(self, device):
        s = self.streams.get(device)
        if s is None:
            s = torch.cuda.Stream(device=device)
            self.streams[device] = s
        return s

    def launch(self, device, fn, keep):
        cur = torch.cuda.current_stream(device)
        side = self.stream(device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            fn()
        self.refs.append(keep)

    def join(self):
        for dev, s in self.streams.items():
            torch.cuda.current_stream(dev).wait_stream(s)
        self.refs.clear()


_ASYNC_WGRAD = _AsyncWgrad()


class prepared_weights:
    """`with prepared_weights():` around forward + backward of one step (hifihr_amd/traineval.forward_backward): the per-step
    weight re-layout launch on entry; `async_wgrad=True` additionally moves weight gradients to a side stream that is joined on
    exit (before the optimizer runs)."""

    def __init__(self, async_wgrad=False):
        self.async_wgrad = bool(async_wgrad) and os.environ.get("HIFIHR_ASYNC_WGRAD", "1") != "0"

    def __enter__(self):
        # branches left over from a forward outside any scope (evaluation): joined now -- or, when this scope opens inside a capture, dropped
        # (torch.cuda.graph synchronises the device before it starts capturing; a capturing stream must not wait on an un-captured one)
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            side_branch._pending.clear()
        else:
            side_branch.join_pending()
        if os.environ.get("HIFIHR_WEIGHT_PREP", "1") != "0":
            _WEIGHT_PREP.begin()
        _DEFER_DW.clear()
        _DEFER_DW.active = True
        # not inside a hipGraph capture: forked branches of a replayed graph ran SLOWER here (8.03 vs 7.74 ms/step) while the same
        # fork in the eager step gains (7.69 vs 7.83)
        _ASYNC_WGRAD.active = self.async_wgrad and (not torch.cuda.is_current_stream_capturing() or os.environ.get("HIFIHR_ASYNC_WGRAD_GRAPH") == "1")
        return self

    def __exit__(self, *exc):
        _WEIGHT_PREP.end()
        _DEFER_DW.active = False
        if exc[0] is None:
            _DEFER_DW.flush()                 # the step's deferred weight-gradient transforms, one launch (in front of the optimizer)
        else:
            _DEFER_DW.clear()
        side_branch.join_pending()
        if _ASYNC_WGRAD.active:
            _ASYNC_WGRAD.active = False
            _ASYNC_WGRAD.join()
        return False


def _wino_conv(lib, x, w_krsc, y, stats, N, H, W, C, K, flip, keep_v=False, bias=None, act=0, U=None, dy_out=None, tile=None, mask=None,
               pair=None):
    """y[N][H][W][K] = conv3x3(x[N][H][W][C], w[K][3][3][C]) through weight / input transform, 16 batched GEMMs, output
    transform (csrc/wino.hip).  flip = 1: w is the [K'][3][3][C'] transpose used by backward-data (rotated filter).
    keep_v: return the transformed input V[16][T][C] in a tensor of its own (the Winograd weight gradient consumes it)."""
    dev = x.device
    # `tile` = (m, positions, tiles) of the LAYER (backward-data runs with C and K swapped but must use the forward's tile edge)
    m, P, T = tile if tile is not None else _wino_tile(lib, N, H, W, C, K)
    prepared = U is not None                          # Winograd-domain weights already computed by the step's weight_prep launch
    if not prepared:
        U = _wino_scratch(dev, "U", P * K * C)
    V = torch.empty(P * T * C, device=dev, dtype=torch.float32) if keep_v else _wino_scratch(dev, "V", P * T * C)
    M = _wino_scratch(dev, "M", P * T * K)
    key = ("wino", N, H, W, C, K, m)
    nb = _CONV_WS_BYTES.get(key)
    if nb is None:
        nb = lib.wino_gemm_workspace_bytes(N, H, W, C, K, m)
        _CONV_WS_BYTES[key] = nb
    ws = None
    if nb:
        ws = _CONV_WS.get(dev)
        if ws is None or ws.numel() * 4 < nb:
            if ws is not None:
                _RETIRED_SCRATCH.append(ws)
            ws = torch.zeros(max(nb, 32 << 20) // 4 + 64, dtype=torch.float32, device=dev)
            _CONV_WS[dev] = ws
    if PROFILE.on:            # (pair: the layer maps K -> C channels in this call's naming; bench.py times the pair launch for it)
        PROFILE.conv_log.append((("wino", N, H, W, K, C, m), "gemm-pair") if pair is not None else (("wino", N, H, W, C, K, m), "gemm"))
    if not prepared:
        lib.wino_weight_transform(w_krsc, U, K, C, flip, m)
    if dy_out is not None:                            # backward: x is dy, the backward-weight transform Y' comes out of the same read
        lib.wino_input_dy_transform(x, V, dy_out, N, H, W, C, m)
    else:
        lib.wino_input_transform(x, V, N, H, W, C, m)
    if pair is not None:
        # backward of a layer that maps K -> C channels in THIS call's naming: its backward-weight product rides in the same launch
        # (pair = (the forward's transformed input, the slab buffer, parts); hifihr_wino4_bwd_gemm_pair)
        vx, dU, parts = pair
        lib.wino4_bwd_gemm_pair(V, U, M, vx, dy_out, dU, N, H, W, K, C, parts)
    else:
        lib.wino_gemm(V, U, M, N, H, W, C, K, ws=ws, m=m)      # csrc/gemm.hip (16x16x4 f32 MFMA); odd channel counts: conv.hip
    lib.wino_output_transform(M, y, stats, N, H, W, K, bias=bias, act=act, m=m, mask=mask)      # mask: y = mask > 0 ? y : 0 (m == 4)
    return V if keep_v else None


def _stem_c3_wgrad(lib, ctx, x, gy):
    """Weight gradient of the 3-channel stem on an NHWC4 image.  The slab kernel writes the parameter's [K][R][S][3] layout itself
    (hifihr_conv2d_bwd_weight_c3: straight into the flat gradient buffer); any other shape goes through a 4-channel temporary.
    -> the gradient to return to autograd (None when it was accumulated in place)."""
    N, H, W, C, K, R, S, stride, pad = ctx.geom
    w, w3 = ctx.w_param, ctx.w3
    Cw = w3.shape[1]
    direct = getattr(w, "_hifihr_direct_grad", False) and w.grad is not None and w.grad.is_contiguous(memory_format=_CL)
    if PROFILE.on:
        PROFILE.conv_log.append(((N, H, W, C, K, R, S, stride, pad), "wgrad"))
    if Cw == 3 and lib.conv2d_bwd_weight_c3_supported(N, H, W, K, R, S, stride, pad):
        tgt = w.grad if direct else torch.zeros_like(w3, memory_format=_CL)
        nslab = lib.conv2d_wgrad_workspace_bytes(N, H, W, C, K, R, S, stride, pad)
        PROFILE.bracket("conv_wgrad", lambda: lib.conv2d_bwd_weight_c3(x, gy, tgt, N, H, W, K, R, S, stride, pad, _wgrad_slabs(gy.device, nslab)))
    else:
        dw4 = torch.zeros((K, C, R, S), device=gy.device).contiguous(memory_format=_CL)
        nslab = lib.conv2d_wgrad_workspace_bytes(N, H, W, C, K, R, S, stride, pad)
        PROFILE.bracket("conv_wgrad", lambda: lib.conv2d_bwd_weight(x, gy, dw4, N, H, W, C, K, R, S, stride, pad,
                                                                     ws=_wgrad_slabs(gy.device, nslab) if nslab else None))
        if not direct:
            return dw4[:, :Cw].contiguous(memory_format=_CL)
        w.grad.add_(dw4[:, :Cw])
    if direct:
        _grad_ready(w)
        return None
    return tgt


class _Conv2dMFMA(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, pad, want_stats=False, bias=None, relu=False, grad_premasked=False, mask_input_grad=False, fork=False):
        """fork (round 4): the input has a SECOND consumer (a residual block's identity branch, the downsample convolution of a stage's first
        block).  The function then returns an alias of x as an extra output; that consumer reads the alias, its gradient comes back to THIS
        backward and is added where the backward-data product is stored (hifihr_conv3x3_c64_wino_res / hifihr_conv2d_bwd_data_pre_res)
        instead of in an elementwise pass of autograd's (5 of them per ResNet-18 step: 52 us).
        grad_premasked (relu=True, frozen bias): the gradient reaches this layer ALREADY multiplied by [y > 0] -- its consumer applied
        this layer's ReLU backward where it produced the gradient (a convolution with mask_input_grad, or a max-pool with relu_input).
        mask_input_grad: x is a ReLU's output; dx is returned multiplied by [x > 0] (fused into the F(4x4, 3x3) output transform of the
        backward-data product, else one bias_relu_bwd pass), so the producer of x can be told grad_premasked.  (VGG19 of the perceptual
        loss: the ReLU backward passes over (dy, y) of its 150-620 MB maps were 1.3 ms per step at batch 48.)"""
        require_cuda(x, w)
        lib = get_lib()
        x = x.contiguous(memory_format=_CL)
        wk = w.contiguous(memory_format=_CL)                      # physical [K][R][S][C]
        ctx.grad_premasked, ctx.mask_input_grad = bool(grad_premasked), bool(mask_input_grad)
        ctx.fork, ctx.want_stats = bool(fork), bool(want_stats)
        assert not (fork and mask_input_grad), "fork and mask_input_grad are not combined"
        ctx.wino_allowed = _wino_allowed()                        # the backward runs outside any conv_precision scope: it follows the forward
        N, C, H, W = x.shape
        K, Cw, R, S = wk.shape
        ctx.w3 = None
        if Cw != C:
            # the 3-channel stem on an NHWC4 image (zero fourth plane): the kernels read a filter zero-padded to 4 channels -- from the
            # step's weight_prep launch (kind 5) when there is one -- and the gradient goes back in the parameter's own 3-channel layout
            assert Cw < C == (Cw + 3) // 4 * 4, (x.shape, w.shape)
            ctx.w3 = wk
            w4 = _WEIGHT_PREP.get(w, wk, 5)
            if w4 is None:
                w4 = torch.zeros(K, R, S, C, device=x.device)
                w4[..., :Cw] = wk.permute(0, 2, 3, 1)
            wk = w4.view(K, R, S, C).permute(0, 3, 1, 2)
        assert C % 4 == 0, (x.shape, w.shape)
        OH, OW = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
        y = torch.empty((N, K, OH, OW), device=x.device, dtype=torch.float32, memory_format=_CL)
        stats = None
        wino = _wino_ok(C, K, R, S, stride, pad) and not (want_stats and (bias is not None or relu))
        v_saved = None
        # 64 -> 64: Winograd with the transforms in registers, when the step's weight_prep launch has U (kind 1) ready
        U64 = _WEIGHT_PREP.get(w, wk, 1) if (Cw == C and _wino2_fused_ok(lib, N, H, W, C, K, R, S, stride, pad, device=x.device)
                                             and not (want_stats and (bias is not None or relu))) else None
        if U64 is not None:
            stats = _ZERO_POOL.acquire(lib.bn_stats_floats(K), x.device) if want_stats else None
            if PROFILE.on:
                PROFILE.conv_log.append(((N, H, W, C, K, R, S, stride, pad), "fwd-wino2"))
            PROFILE.bracket("conv_fwd", lambda: lib.conv3x3_c64_wino(x, U64, bias, relu, y, stats, N, H, W))
        elif wino:
            stats = _ZERO_POOL.acquire(lib.bn_stats_floats(K), x.device) if want_stats else None
            keep = bool(ctx.needs_input_grad[1])
            box = []
            tile = _wino_tile(lib, N, H, W, C, K)
            U = _WEIGHT_PREP.get(w, wk, 1 if tile[0] == 2 else 3)
            PROFILE.bracket("conv_fwd_wino", lambda: box.append(_wino_conv(lib, x, wk, y, stats, N, H, W, C, K, 0, keep_v=keep,
                                                                            bias=bias, act=1 if relu else 0, U=U, tile=tile)))
            v_saved = box[0]
        elif want_stats:       # per-channel sum / sum of squares of y from the conv epilogue, for the batch-norm that follows
            stats = _ZERO_POOL.acquire(lib.bn_stats_floats(K), x.device)
            ws = _conv_ws(lib, x.device, (N, H, W, C, K, R, S, stride, pad), False)
            PROFILE.bracket("conv_fwd", lambda: lib.conv2d_fwd_bnstats(x, wk, y, stats, N, H, W, C, K, R, S, stride, pad, ws=ws))
        else:
            ws = _conv_ws(lib, x.device, (N, H, W, C, K, R, S, stride, pad), False) if (bias is None and not relu) else None
            PROFILE.bracket("conv_fwd", lambda: lib.conv2d_fwd(x, wk, bias, y, N, H, W, C, K, R, S, stride, pad, ws=ws,
                                                              act=1 if relu else 0))
        if PROFILE.on and not wino and U64 is None:
            PROFILE.conv_log.append(((N, H, W, C, K, R, S, stride, pad), "fwd"))
        ctx.geom = (N, H, W, C, K, R, S, stride, pad)
        # Winograd layers keep the transformed input V (4x the size of x, 288 GB of HBM do not care) instead of x: the weight
        # gradient reduces Y' . V in the transform domain and backward-data needs neither
        ctx.save_for_backward(x if (v_saved is None or mask_input_grad) else None, wk, y if (relu and not grad_premasked) else None, v_saved)
        ctx.w_param, ctx.b_param, ctx.relu = w, bias, relu
        ctx.set_materialize_grads(False)         # no zero-fill launch for the (non-differentiable) stats output
        outs = [y]
        if want_stats:
            ctx.mark_non_differentiable(stats)
            outs.append(stats)
        if fork:
            outs.append(x.view_as(x))
        return tuple(outs) if len(outs) > 1 else y

    @staticmethod
    def backward(ctx, gy, *rest):
        x, wk, y, v_saved = ctx.saved_tensors
        lib = get_lib()
        N, H, W, C, K, R, S, stride, pad = ctx.geom
        g_fork = rest[-1] if (getattr(ctx, "fork", False) and rest) else None      # gradient of the input's other consumer (None: it had none; _CtxShim: no fork)
        if g_fork is not None:
            g_fork = g_fork.contiguous(memory_format=_CL)
        if gy is None:
            return (g_fork,) + (None,) * 9
        gy = gy.contiguous(memory_format=_CL)
        dx = dw = db_ret = Yt_done = pair_done = None
        c64_pair_done = False
        if ctx.relu or ctx.b_param is not None:
            # conv + bias (+ ReLU) epilogue: masked gradient and the bias gradient in one small launch
            b = ctx.b_param
            db_t, db_ret = _acc_target(b, b.shape, gy.device) if (b is not None and ctx.needs_input_grad[5]) else (None, None)
            if ctx.relu and ctx.grad_premasked:
                assert db_t is None, "grad_premasked needs a frozen bias (the bias gradient comes out of the masking pass)"
            elif ctx.relu:
                g = torch.empty_like(gy, memory_format=_CL)
                M = gy.numel() // K
                PROFILE.bracket("bias_relu_bwd", lambda: lib.bias_relu_bwd(gy, y, M, K, g, db_t))
                gy = g
            elif db_t is not None:
                raise NotImplementedError("gradient of a conv bias without ReLU")   # only the frozen VGG19 has such a layer
            if b is not None and db_t is not None and db_ret is None:
                _grad_ready(b)
        masked = False
        if ctx.needs_input_grad[0] and _wino_ok(C, K, R, S, stride, pad, ctx.wino_allowed):
            # backward-data of a stride-1 3x3 = the same Winograd pipeline on dy with the transposed, rotated filter
            dx = torch.empty((N, C, H, W), device=gy.device, dtype=torch.float32, memory_format=_CL)
            tile = _wino_tile(lib, N, H, W, C, K)
            wm, wP, wT = tile
            mk = x if (ctx.mask_input_grad and wm == 4) else None         # [x > 0] applied in the output transform
            masked = mk is not None
            U2 = _WEIGHT_PREP.get(ctx.w_param, wk, 2 if wm == 2 else 4)
            if ctx.needs_input_grad[1] and v_saved is not None:      # the Winograd backward-weight below wants A dy A^T: same read of dy
                # a side-stream weight gradient reads it while the next layer's backward-data already runs: a buffer of its own
                Yt_done = _wino_scratch(gy.device, ("Yt", ctx.w_param.data_ptr()) if _ASYNC_WGRAD.active else "Yt", wP * wT * K)
            # both gradients wanted, F(4x4), prepared filter, weight gradient on THIS stream: the two products share one launch
            if (Yt_done is not None and wm == 4 and U2 is not None and _GEMM_PAIR and min(C, K) % 64 == 0
                    and lib.wino4_bwd_gemm_pair_supported(N, H, W, C, K)):
                pparts = lib.wino_wgrad_parts(N, H, W, C, K, wm)
                if pparts > 0:
                    # (a side-stream weight-gradient transform reads the slabs while the next layer's pair already runs: a buffer of its own)
                    own = _ASYNC_WGRAD.active or _DEFER_DW.wants(ctx.w_param, _direct_grad(ctx.w_param) and ctx.w_param.grad.is_contiguous(memory_format=_CL))
                    pair_done = (v_saved, _wino_scratch(gy.device, ("dUp", ctx.w_param.data_ptr()) if own else "dUp",
                                                        pparts * wP * K * C), pparts)

            def run():
                if U2 is not None:
                    _wino_conv(lib, gy, None, dx, None, N, H, W, K, C, 1, U=U2, dy_out=Yt_done, tile=tile, mask=mk, pair=pair_done)
                else:
                    wt = _wino_scratch(gy.device, "wt", wk.numel())
                    lib.weight_transpose(wk, wt, K, R * S, C)
                    _wino_conv(lib, gy, wt, dx, None, N, H, W, K, C, 1, dy_out=Yt_done, tile=tile, mask=mk)
            PROFILE.bracket("conv_dgrad_wino", run)
            if g_fork is not None:
                dx = dx + g_fork
        elif ctx.needs_input_grad[0] and ctx.w3 is None and _wino2_fused_ok(lib, N, H, W, C, K, R, S, stride, pad, ctx.wino_allowed, device=gy.device) and \
                _WEIGHT_PREP.get(ctx.w_param, wk, 2) is not None:
            # 64 -> 64: the same one-launch Winograd kernel on dy with U' (kind 2: transposed, rotated filter)
            dx = torch.empty_like(x, memory_format=_CL)
            U2 = _WEIGHT_PREP.get(ctx.w_param, wk, 2)
            if ctx.needs_input_grad[1] and v_saved is None and _C64_PAIR and lib.conv3x3_c64_bwd_pair_supported(N, H, W):
                # both gradients wanted: the data gradient and the pixel-reduction weight gradient share ONE launch (csrc/conv_halo.hip
                # conv_c64_bwd_pair_kernel; on this stream also when weight gradients otherwise go to the side stream)
                w = ctx.w_param
                tgt = w.grad if (getattr(w, "_hifihr_direct_grad", False) and w.grad is not None
                                 and w.grad.is_contiguous(memory_format=_CL)) else None
                if tgt is None:
                    dw = torch.zeros_like(wk, memory_format=_CL)
                    tgt = dw
                if PROFILE.on:
                    PROFILE.conv_log.append(((N, H, W, C, K, R, S, stride, pad), "c64-pair"))
                nslab = lib.conv2d_wgrad_workspace_bytes(N, H, W, C, K, R, S, stride, pad)
                if nslab and dw is None and _DEFER_DW.wants(w, True):
                    # the slab sum joins the step's deferred launches (one per step in front of the optimizer): slabs in a buffer of the layer's own
                    own = _wino_scratch(gy.device, ("c64slabs", w.data_ptr()), (nslab + 3) // 4)
                    box = []
                    PROFILE.bracket("conv_dgrad", lambda: box.append(lib.conv3x3_c64_bwd_pair_slabs(gy, U2, g_fork, dx, x, own, N, H, W)))
                    _DEFER_DW.add_halo(own, box[-1], tgt)
                else:
                    slabs = _wgrad_slabs(gy.device, nslab) if nslab else None
                    PROFILE.bracket("conv_dgrad", lambda: lib.conv3x3_c64_bwd_pair(gy, U2, g_fork, dx, x, tgt, N, H, W, ws=slabs))
                c64_pair_done = True
            elif PROFILE.on:
                PROFILE.conv_log.append(((N, H, W, C, K, R, S, stride, pad), "dgrad-wino2"))
            if c64_pair_done:
                pass
            elif g_fork is not None:
                PROFILE.bracket("conv_dgrad", lambda: lib.conv3x3_c64_wino_res(gy, U2, g_fork, dx, N, H, W))
            else:
                PROFILE.bracket("conv_dgrad", lambda: lib.conv3x3_c64_wino(gy, U2, None, False, dx, None, N, H, W))
        elif ctx.needs_input_grad[0]:
            dx = torch.empty_like(x, memory_format=_CL)
            ws = _conv_ws(lib, x.device, (N, H, W, C, K, R, S, stride, pad), True)
            if PROFILE.on:
                PROFILE.conv_log.append(((N, H, W, C, K, R, S, stride, pad), "dgrad"))
            wt = _WEIGHT_PREP.get(ctx.w_param, wk, 0)
            plus = getattr(ctx, "plus1x1", None)          # (dy, transposed filter) of a 1x1 convolution of the same input (_Conv2dPair.backward)
            if plus is not None:
                assert wt is not None and g_fork is None
                if PROFILE.on:
                    PROFILE.conv_log[-1] = ((N, H, W, C, K, R, S, stride, pad), "dgrad+1x1")
                PROFILE.bracket("conv_dgrad", lambda: lib.conv2d_bwd_data_pre_plus1x1(gy, wt, plus[0], plus[1], dx, N, H, W, C, K, R, S, stride, pad))
            elif wt is not None and g_fork is not None:
                PROFILE.bracket("conv_dgrad", lambda: lib.conv2d_bwd_data_pre_res(gy, wt, g_fork, dx, N, H, W, C, K, R, S, stride, pad, ws=ws))
            elif wt is not None:                   # [C][R][S][K] transpose from the step's weight_prep launch
                PROFILE.bracket("conv_dgrad", lambda: lib.conv2d_bwd_data_pre(gy, wt, dx, N, H, W, C, K, R, S, stride, pad, ws=ws))
            else:
                scratch = torch.empty(wk.numel(), device=x.device, dtype=torch.float32)
                PROFILE.bracket("conv_dgrad", lambda: lib.conv2d_bwd_data(gy, wk, dx, scratch, N, H, W, C, K, R, S, stride, pad, ws=ws))
                if g_fork is not None:
                    dx = dx + g_fork
        if ctx.mask_input_grad and dx is not None and not masked:
            dxm = torch.empty_like(dx, memory_format=_CL)
            PROFILE.bracket("bias_relu_bwd", lambda: lib.bias_relu_bwd(dx, x, dx.numel() // C, C, dxm, None))
            dx = dxm
        if ctx.needs_input_grad[1] and ctx.w3 is not None:
            dw = _stem_c3_wgrad(lib, ctx, x, gy)
        elif c64_pair_done:
            if dw is None:
                _grad_ready(ctx.w_param)
        elif ctx.needs_input_grad[1]:
            w = ctx.w_param
            tgt = w.grad if (getattr(w, "_hifihr_direct_grad", False) and w.grad is not None
                             and w.grad.is_contiguous(memory_format=_CL)) else None
            if tgt is None:
                dw = torch.zeros_like(wk, memory_format=_CL)
                tgt = dw
            # accumulates (fp32 atomics) straight into the flat gradient buffer when the parameter lives in one
            if v_saved is not None:
                wm, wP, T = _wino_tile(lib, N, H, W, C, K)
                Yt = Yt_done if Yt_done is not None else _wino_scratch(
                    gy.device, ("Yt", w.data_ptr()) if _ASYNC_WGRAD.active else "Yt", wP * T * K)
                parts = lib.wino_wgrad_parts(N, H, W, C, K, wm)  # > 0: the slab form on csrc/gemm.hip (no atomics, nothing to zero)
                defer = wm == 4 and parts > 0 and _DEFER_DW.wants(w, dw is None)
                if parts > 0:
                    dU = _wino_scratch(gy.device, ("dUp", w.data_ptr()) if (_ASYNC_WGRAD.active or defer) else "dUp", parts * wP * K * C)
                else:
                    key = (gy.device, "dU", 16 * K * C)
                    dU = _WINO_SCRATCH.get(key)
                    if dU is None:                               # zero-initialised once; wino_dw_transform hands it back zeroed
                        dU = torch.zeros(16 * K * C, device=gy.device, dtype=torch.float32)
                        _WINO_SCRATCH[key] = dU

                if PROFILE.on and pair_done is None:
                    PROFILE.conv_log.append((("wino", N, H, W, C, K, wm), "gemm-tn"))

                def run_w():
                    if Yt_done is None:
                        lib.wino_dy_transform(gy, Yt, N, H, W, K, wm)
                    if pair_done is not None:              # the product already ran beside backward-data (same slab buffer)
                        if defer:
                            _DEFER_DW.add(pair_done[1], pair_done[2], tgt, K, C)
                        else:
                            lib.wino_dw_transform_parts(pair_done[1], pair_done[2], tgt, K, C, wm)
                    elif parts > 0:
                        lib.wino_wgrad_gemm_parts(v_saved, Yt, dU, N, H, W, C, K, parts, wm)
                        if defer:
                            _DEFER_DW.add(dU, parts, tgt, K, C)
                        else:
                            lib.wino_dw_transform_parts(dU, parts, tgt, K, C, wm)
                    else:
                        lib.wino_wgrad_gemm(v_saved, Yt, dU, N, H, W, C, K)
                        lib.wino_dw_transform(dU, tgt, K, C, clear=True)
                go = lambda: PROFILE.bracket("conv_wgrad_wino", run_w)
                keep = (gy, v_saved, Yt, tgt)
            else:
                if PROFILE.on:
                    PROFILE.conv_log.append(((N, H, W, C, K, R, S, stride, pad), "wgrad"))
                nslab = lib.conv2d_wgrad_workspace_bytes(N, H, W, C, K, R, S, stride, pad)
                go = lambda: PROFILE.bracket("conv_wgrad", lambda: lib.conv2d_bwd_weight(
                    x, gy, tgt, N, H, W, C, K, R, S, stride, pad, ws=_wgrad_slabs(gy.device, nslab) if nslab else None))
                keep = (gy, x, tgt)
            if _ASYNC_WGRAD.active and dw is None:
                _ASYNC_WGRAD.launch(gy.device, go, keep)
            else:
                go()
            if dw is None:
                _grad_ready(w)
        if dx is None and g_fork is not None:
            dx = g_fork
        return dx, dw, None, None, None, db_ret, None, None, None, None


def conv2d_bias_act(x, w, bias, stride=1, pad=0, relu=True, grad_premasked=False, mask_input_grad=False):
    """F.conv2d(x, w, bias, stride, pad) followed by ReLU if `relu`, in one launch (LightEstimator, reference
    network/res_encoder.py:150-210; VGG19 features of the perceptual loss, utils/perceptual_loss.py:27-36).
    grad_premasked / mask_input_grad: where the ReLU backward of a conv + ReLU chain runs (see _Conv2dMFMA.forward)."""
    return _Conv2dMFMA.apply(x, w, stride, pad, False, bias, relu, grad_premasked, mask_input_grad)


def conv2d_bias_relu(x, w, bias, stride=1, pad=0):
    return conv2d_bias_act(x, w, bias, stride, pad, True)


def conv2d(x, w, stride=1, pad=0, want_stats=False, fork=False):
    """F.conv2d(x, w, None, stride, pad) for channels_last fp32 tensors (reference network/res_encoder.py:364-373).
    want_stats=True additionally returns the [2,K] (sum, sum of squares) of the output for `bn_act`; fork=True an alias of x as the
    LAST output, for x's other consumer (see _Conv2dMFMA.forward)."""
    return _Conv2dMFMA.apply(x, w, stride, pad, want_stats, None, False, False, False, fork)


class _Conv2dPair(torch.autograd.Function):
    """y1 = conv(x, w1; 3x3, pad 1), y2 = conv(x, w2; 1x1, pad 0), both with `stride` and their batch-norm statistics, in ONE launch
    (hifihr_conv2d_fwd_bnstats_pair): conv1 and downsample[0] of a residual stage's first block read the same x.  The backward runs the two
    convolutions' own backward passes (_Conv2dMFMA.backward on shims): the downsample branch's first, its data gradient then enters the 3x3
    convolution's backward-data launch as the `fork` residual -- exactly what the two separate functions did."""

    @staticmethod
    def forward(ctx, x, w1, w2, stride):
        require_cuda(x, w1, w2)
        lib = get_lib()
        x = x.contiguous(memory_format=_CL)
        wk1, wk2 = w1.contiguous(memory_format=_CL), w2.contiguous(memory_format=_CL)
        N, C, H, W = x.shape
        K1, K2 = wk1.shape[0], wk2.shape[0]
        OH, OW = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
        y1 = torch.empty((N, K1, OH, OW), device=x.device, dtype=torch.float32, memory_format=_CL)
        y2 = torch.empty((N, K2, OH, OW), device=x.device, dtype=torch.float32, memory_format=_CL)
        st1 = _ZERO_POOL.acquire(lib.bn_stats_floats(K1), x.device)
        st2 = _ZERO_POOL.acquire(lib.bn_stats_floats(K2), x.device)
        if PROFILE.on:
            PROFILE.conv_log.append((("pair", N, H, W, C, K1, K2, stride), "fwd-pair"))
        PROFILE.bracket("conv_fwd", lambda: lib.conv2d_fwd_bnstats_pair(x, wk1, y1, st1, K1, 3, 1, wk2, y2, st2, K2, 1, 0, N, H, W, C, stride))
        ctx.geom1, ctx.geom2 = (N, H, W, C, K1, 3, 3, stride, 1), (N, H, W, C, K2, 1, 1, stride, 0)
        ctx.save_for_backward(x, wk1, wk2)
        ctx.w1, ctx.w2 = w1, w2
        ctx.wino_allowed = _wino_allowed()
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(st1, st2)
        return y1, st1, y2, st2

    @staticmethod
    def backward(ctx, gy1, _gs1, gy2, _gs2):
        x, wk1, wk2 = ctx.saved_tensors
        need_x, need_w1, need_w2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]

        def shim(wk, geom, w, need_w, fork):
            return _CtxShim((x, wk, None, None), geom=geom, w_param=w, b_param=None, relu=False, w3=None, wino_allowed=ctx.wino_allowed,
                            grad_premasked=False, mask_input_grad=False, fork=fork,
                            needs_input_grad=(need_x, need_w, False, False, False, False, False, False, False, False))
        dx2 = dw2 = dx = dw1 = None
        # Both gradients present and the step's re-laid-out filters at hand: the 1x1 convolution's data gradient -- it lands on the pixels
        # (stride i, stride j) only -- rides in the 3x3 convolution's backward-data launch as one more tap of that parity class
        # (hifihr_conv2d_bwd_data_pre_plus1x1), instead of a launch of its own whose mostly-zero result comes back in as the fork residual.
        N, H, W, C, K1, R, S, stride, pad = ctx.geom1
        plus = None
        if gy1 is not None and gy2 is not None and need_x and wk2.shape[0] == K1:
            lib = get_lib()
            if lib.conv2d_bwd_data_pre_plus1x1_supported(N, H, W, C, K1, R, S, stride, pad):
                wt1, wt2 = _WEIGHT_PREP.get(ctx.w1, wk1, 0), _WEIGHT_PREP.get(ctx.w2, wk2, 0)
                if wt1 is not None and wt2 is not None:
                    plus = (gy2.contiguous(memory_format=_CL), wt2)
        if plus is not None:
            # the two weight gradients likewise: the 1x1 convolution's patch column is tap (pad, pad) of the 3x3's -- its column tiles join the
            # 3x3's launch (hifihr_conv2d_bwd_weight_plus1x1) unless weight gradients run on the side stream
            w_both = (need_w1 and need_w2 and not _ASYNC_WGRAD.active
                      and lib.conv2d_bwd_weight_plus1x1_supported(N, H, W, C, K1, R, S, stride, pad))
            if w_both:
                gy1c = gy1.contiguous(memory_format=_CL)

                def target(w, wk):
                    if getattr(w, "_hifihr_direct_grad", False) and w.grad is not None and w.grad.is_contiguous(memory_format=_CL):
                        return w.grad, None
                    t = torch.zeros_like(wk, memory_format=_CL)
                    return t, t
                t1, dw1 = target(ctx.w1, wk1)
                t2, dw2 = target(ctx.w2, wk2)
                if PROFILE.on:
                    PROFILE.conv_log.append(((N, H, W, C, K1, R, S, stride, pad), "wgrad+1x1"))
                PROFILE.bracket("conv_wgrad", lambda: lib.conv2d_bwd_weight_plus1x1(x, gy1c, t1, plus[0], t2, N, H, W, C, K1, R, S, stride, pad))
                if dw1 is None:
                    _grad_ready(ctx.w1)
                if dw2 is None:
                    _grad_ready(ctx.w2)
            else:
                s2 = shim(wk2, ctx.geom2, ctx.w2, need_w2, False)
                s2.needs_input_grad = (False,) + tuple(s2.needs_input_grad[1:])      # its weight gradient only
                dw2 = _Conv2dMFMA.backward(s2, gy2)[1]
            s1 = shim(wk1, ctx.geom1, ctx.w1, need_w1 and not w_both, False)
            s1.plus1x1 = plus
            dx, dw1b = _Conv2dMFMA.backward(s1, gy1)[:2]
            return dx, (dw1 if w_both else dw1b), dw2, None
        if gy2 is not None:
            dx2, dw2 = _Conv2dMFMA.backward(shim(wk2, ctx.geom2, ctx.w2, need_w2, False), gy2)[:2]
        if gy1 is not None:
            dx, dw1 = _Conv2dMFMA.backward(shim(wk1, ctx.geom1, ctx.w1, need_w1, True), gy1, dx2)[:2]      # (dx2 enters as the fork residual)
        else:
            dx = dx2
        return dx, dw1, dw2, None


def conv2d_pair_ok(x, w1, w2, stride):
    """conv1 (3x3, pad 1) and downsample[0] (1x1, pad 0) of a stage's first block can share ONE forward launch (training mode: both feed a
    batch-norm that wants statistics).  HIFIHR_CONV_ROWS_PAIR=0 (read by the library) keeps the two launches."""
    if not (x.is_cuda and torch.is_tensor(x) and w1.shape[2:] == (3, 3) and w2.shape[2:] == (1, 1) and w1.shape[1] == x.shape[1] == w2.shape[1]):
        return False
    lib = get_lib()
    N, C, H, W = x.shape
    return lib.zero_page_ready(x.device) and lib.conv2d_fwd_bnstats_pair_supported(N, H, W, C, stride, w1.shape[0], 3, 1, w2.shape[0], 1, 0)


def conv2d_pair(x, w1, w2, stride):
    """-> (conv(x, w1), its batch statistics, conv(x, w2), its batch statistics)"""
    return _Conv2dPair.apply(x, w1, w2, stride)


def conv_fork_enabled():
    """HIFIHR_CONV_FORK=0: residual blocks leave the sum of their input's two gradients to autograd (the A/B switch of `fork`)."""
    return os.environ.get("HIFIHR_CONV_FORK", "1") != "0"


class _BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, stats, gamma, beta, residual, act, eps, momentum, running_mean, running_var):
        require_cuda(x, stats, gamma, beta)
        lib = get_lib()
        x = x.contiguous(memory_format=_CL)
        N, C, H, W = x.shape
        M = N * H * W
        res = residual.contiguous(memory_format=_CL) if residual is not None else None
        y = torch.empty_like(x, memory_format=_CL)
        save_mean = torch.empty(C, device=x.device)
        save_invstd = torch.empty(C, device=x.device)
        PROFILE.bracket("bn_fwd", lambda: lib.bn_act_fwd(x, stats, gamma, beta, res, act, M, C, eps, momentum, y, save_mean,
                                                         save_invstd, running_mean, running_var))
        _ZERO_POOL.release(stats)                # consumed and zeroed by the kernel
        # ReLU without a residual input: the backward recomputes the mask from x (csrc/bn.hip masked_grad), y is not kept for it
        ctx.save_for_backward(x, y if (act == 1 and residual is not None) else x.new_empty(0), gamma, beta, save_mean, save_invstd)
        ctx.act, ctx.has_res, ctx.M, ctx.C = act, residual is not None, M, C
        ctx.gamma_param, ctx.beta_param = gamma, beta
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, beta, save_mean, save_invstd = ctx.saved_tensors
        y = y if (ctx.act == 1 and ctx.has_res) else None
        lib = get_lib()
        dy = dy.contiguous(memory_format=_CL)
        dx = torch.empty_like(x, memory_format=_CL)
        dres = torch.empty_like(x, memory_format=_CL) if ctx.has_res else None
        red = _ZERO_POOL.acquire(lib.bn_stats_floats(ctx.C), x.device)

        def acc_target(p):
            if getattr(p, "_hifihr_direct_grad", False) and p.grad is not None:
                return p.grad, None                 # accumulate straight into the flat gradient buffer
            t = torch.zeros(ctx.C, device=x.device)
            return t, t
        dg_t, dg_ret = acc_target(ctx.gamma_param)
        db_t, db_ret = acc_target(ctx.beta_param)
        PROFILE.bracket("bn_bwd", lambda: lib.bn_act_bwd(dy, y, x, save_mean, save_invstd, gamma, beta, ctx.act, ctx.M, ctx.C, red,
                                                         dx, dres, dg_t, db_t))
        _ZERO_POOL.release(red)
        if dg_ret is None:
            _grad_ready(ctx.gamma_param)
        if db_ret is None:
            _grad_ready(ctx.beta_param)
        return dx, None, dg_ret, db_ret, dres, None, None, None, None, None


class _BNActEval(torch.autograd.Function):
    """act(batch_norm(x; running statistics) + residual): evaluation mode (reference train_hrnet.py:119-161 runs model.eval()).
    Forward: one launch of the fused apply kernel (hifihr_bn_act_eval).  Backward (a frozen-statistics layer inside a graph that
    still needs gradients -- not on the training hot path): per-channel affine map, a handful of elementwise torch ops."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, rmean, rvar, act, eps):
        require_cuda(x, gamma, beta)
        N, C, H, W = x.shape
        y = torch.empty_like(x, memory_format=_CL)
        get_lib().bn_act_eval(x, rmean, rvar, gamma, beta, residual, act, N * H * W, C, eps, y)
        ctx.save_for_backward(x, y, gamma, beta, rmean, rvar)
        ctx.act, ctx.eps, ctx.has_res = act, eps, residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, beta, rmean, rvar = ctx.saved_tensors
        invstd = torch.rsqrt(rvar + ctx.eps).view(1, -1, 1, 1)
        sc = gamma.view(1, -1, 1, 1) * invstd
        if ctx.act == 1:
            g = dy * (y > 0)
        elif ctx.act == 2:
            z = (x - rmean.view(1, -1, 1, 1)) * sc + beta.view(1, -1, 1, 1)
            sg = torch.sigmoid(z)
            g = dy * (sg * (1 + z * (1 - sg)))
        else:
            g = dy
        xhat = (x - rmean.view(1, -1, 1, 1)) * invstd
        return g * sc, (g * xhat).sum((0, 2, 3)), g.sum((0, 2, 3)), (g if ctx.has_res else None), None, None, None, None


class _CtxShim:
    """What _Conv2dMFMA.backward / _BNAct.backward read from their autograd context, for the fused functions that run those backward
    passes on tensors they saved themselves."""

    def __init__(self, saved, **kw):
        self.saved_tensors = saved
        self.__dict__.update(kw)


def bn_wino_fusable(x, w, bn, stride, pad):
    """A train-mode BatchNorm2d + ReLU in front of this convolution can run inside its Winograd input transform (csrc/wino4_bn.hip)."""
    if os.environ.get("HIFIHR_BN_WINO_FUSE", "1") == "0" or not bn.training or not x.is_cuda:
        return False
    N, C, H, W = x.shape
    K, Cw, R, S = w.shape
    if Cw != C or not _wino_ok(C, K, R, S, stride, pad):
        return False
    lib = get_lib()
    return (_wino_tile(lib, N, H, W, C, K)[0] == 4 and lib.wino_bn_input_supported(C, 4) and lib.wino_bn_input_supported(K, 4)
            and lib.wino_wgrad_parts(N, H, W, C, K, 4) > 0)


class _WinoLink:
    """Hand-off between the backward passes of two fused functions, attached to the raw convolution output y that the first produced
    and the second consumed: when the consumer's backward has the batch-norm's masked gradient g and its reduction sums, it does NOT
    apply the batch-norm backward; it returns g in place of d loss / d y and leaves the sums here, and the producer's backward applies
    dy = gamma invstd (g - mean g - xhat mean(g xhat)) inside its dual input transform (hifihr_wino_bn_bwd_dual_transform).  Valid only
    because y has exactly one consumer (the fused function) -- BasicBlock guarantees that."""
    __slots__ = ("lazy", "red", "save_mean", "save_invstd", "gamma", "gamma_param", "beta_param", "g")

    def __init__(self):
        self.lazy, self.red, self.g = False, None, None


def _direct_grad(p):
    return getattr(p, "_hifihr_direct_grad", False) and p.grad is not None


def _wino_gemm_ws(lib, dev, N, H, W, C, K, m):
    key = ("wino", N, H, W, C, K, m)
    nb = _CONV_WS_BYTES.get(key)
    if nb is None:
        nb = lib.wino_gemm_workspace_bytes(N, H, W, C, K, m)
        _CONV_WS_BYTES[key] = nb
    if not nb:
        return None
    ws = _CONV_WS.get(dev)
    if ws is None or ws.numel() * 4 < nb:
        if ws is not None:
            _RETIRED_SCRATCH.append(ws)
        ws = torch.zeros(max(nb, 32 << 20) // 4 + 64, dtype=torch.float32, device=dev)
        _CONV_WS[dev] = ws
    return ws


class _BNActWinoConv(torch.autograd.Function):
    """conv3x3(relu(bn(x; batch statistics) (+ residual)), w) with the batch-norm applied inside the Winograd F(4x4, 3x3) input transform
    (hifihr_wino_bn_input_transform): the activation is never written on its own.  With a residual the block output
    relu(bn(x) + residual) is also returned (the next block's identity branch and shortcut convolution read it).
    Backward: ReLU mask, identity-branch gradient and the batch-norm reduction in the epilogue of the backward-data output transform
    (hifihr_wino_output_transform_bnred); the batch-norm apply inside the PRODUCER's dual transform when x came from another fused
    function (_WinoLink), else one hifihr_bn_bwd_apply launch.
    Replaces, per BasicBlock of the reference's ResNet (network/res_encoder.py:364-373, vendored resnet.py BasicBlock.forward), the
    dispatches bn1 -> relu -> conv2 and bn2 -> += identity -> relu -> conv1 of the next block, and their autograd.
    -> (y, stats of y or None, block output or None)."""

    @staticmethod
    def forward(ctx, x, stats, gamma, beta, residual, w, eps, momentum, running_mean, running_var, want_stats, in_link, link):
        require_cuda(x, stats, gamma, beta, w)
        lib = get_lib()
        x = x.contiguous(memory_format=_CL)
        wk = w.contiguous(memory_format=_CL)
        N, C, H, W = x.shape
        K = wk.shape[0]
        dev = x.device
        tile = _wino_tile(lib, N, H, W, C, K)
        m, P, T = tile
        assert m == 4
        res = residual.contiguous(memory_format=_CL) if residual is not None else None
        out = torch.empty_like(x, memory_format=_CL) if res is not None else None
        save_mean, save_invstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
        U = _WEIGHT_PREP.get(w, wk, 3)
        V = torch.empty(P * T * C, device=dev, dtype=torch.float32)       # kept: the Winograd weight gradient reduces Y' . V
        y = torch.empty((N, K, H, W), device=dev, dtype=torch.float32, memory_format=_CL)
        stats_next = _ZERO_POOL.acquire(lib.bn_stats_floats(K), dev) if want_stats else None

        def run():
            Uu = U
            if Uu is None:
                Uu = _wino_scratch(dev, "U", P * K * C)
                lib.wino_weight_transform(wk, Uu, K, C, 0, m)
            lib.wino_bn_input_transform(x, stats, gamma, beta, res, out, V, N, H, W, C, m, eps, momentum, save_mean, save_invstd,
                                        running_mean, running_var)
            M = _wino_scratch(dev, "M", P * T * K)
            lib.wino_gemm(V, Uu, M, N, H, W, C, K, ws=_wino_gemm_ws(lib, dev, N, H, W, C, K, m), m=m)
            lib.wino_output_transform(M, y, stats_next, N, H, W, K, m=m)
        if PROFILE.on:
            PROFILE.conv_log.append((("wino", N, H, W, C, K, m), "gemm"))
        PROFILE.bracket("bn_conv_fwd_wino", run)
        _ZERO_POOL.release(stats)                # consumed and zeroed by the fused transform
        ctx.save_for_backward(x, out if out is not None else x.new_empty(0), gamma, beta, save_mean, save_invstd, wk, V, y)
        ctx.geom = (N, H, W, C, K, 3, 3, 1, 1)
        ctx.tile = tile
        ctx.has_res = res is not None
        ctx.w_param, ctx.gamma_param, ctx.beta_param = w, gamma, beta
        ctx.in_link, ctx.link = in_link, link
        if in_link is not None:
            in_link.lazy = True                  # our backward may hand the batch-norm apply over to the producer of x
        ctx.set_materialize_grads(False)
        if stats_next is not None:
            ctx.mark_non_differentiable(stats_next)
        return y, stats_next, out

    @staticmethod
    def backward(ctx, gy, _gstats, g_out):
        x, out, gamma, beta, save_mean, save_invstd, wk, V, y = ctx.saved_tensors
        N, H, W, C, K = ctx.geom[:5]
        if gy is None:                            # the convolution's output went nowhere: only the identity branch brings a gradient
            raise NotImplementedError("fused batch-norm + Winograd convolution whose convolution output is unused")
        lib = get_lib()
        dev = x.device
        m, P, T = ctx.tile
        w = ctx.w_param
        need = ctx.needs_input_grad
        gy = gy.contiguous(memory_format=_CL)
        link, in_link = ctx.link, ctx.in_link
        if os.environ.get("HIFIHR_BN_WINO_BWD", "1") == "0":      # A/B: the unfused backward (Winograd pipeline, add, batch-norm backward)
            conv = _CtxShim((None, wk, None, V), geom=ctx.geom, w_param=w, b_param=None, relu=False, w3=None, wino_allowed=True,
                            grad_premasked=False, mask_input_grad=False,
                            needs_input_grad=(True, need[5], False, False, False, False, False))
            d_a, dw = _Conv2dMFMA.backward(conv, gy)[:2]
            if g_out is not None:
                d_a = d_a + g_out
            bn = _CtxShim((x, out if ctx.has_res else x.new_empty(0), gamma, beta, save_mean, save_invstd), act=1, has_res=ctx.has_res,
                          M=N * H * W, C=C, gamma_param=ctx.gamma_param, beta_param=ctx.beta_param)
            dx, _, dg, db, dres = _BNAct.backward(bn, d_a)[:5]
            return dx, None, dg, db, dres, dw, None, None, None, None, None, None, None

        def acc_target(p, n):
            if _direct_grad(p):
                return p.grad, None
            t = torch.zeros(n, device=dev)
            return t, t
        res_out = out if ctx.has_res else None
        gadd = g_out.contiguous(memory_format=_CL) if g_out is not None else None
        g_in = torch.empty_like(x, memory_format=_CL)
        red = _ZERO_POOL.acquire(lib.bn_stats_floats(C), dev)
        lazy_in = link.lazy and link.red is not None            # gy is the masked gradient of the NEXT batch-norm, un-applied
        dw_box = [None]
        direct_w = _direct_grad(w) and w.grad.is_contiguous(memory_format=_CL)

        def run():
            # 1. transforms of d loss / d y: V' (backward-data) and Y' (backward-weight)
            V2 = _wino_scratch(dev, "V", P * T * K)
            Yt = _wino_scratch(dev, "Yt", P * T * K)
            if lazy_in:
                lib.wino_bn_bwd_dual_transform(gy, y, link.save_mean, link.save_invstd, link.gamma, link.red, V2, Yt, N, H, W, K, m,
                                               link.gamma_param.grad, link.beta_param.grad)
            else:
                lib.wino_input_dy_transform(gy, V2, Yt, N, H, W, K, m)
            # 2. backward-data product on the rotated transposed filter
            U2 = _WEIGHT_PREP.get(w, wk, 4)
            if U2 is None:
                wt = _wino_scratch(dev, "wt", wk.numel())
                lib.weight_transpose(wk, wt, K, 9, C)
                U2 = _wino_scratch(dev, "U", P * K * C)
                lib.wino_weight_transform(wt, U2, C, K, 1, m)
            M2 = _wino_scratch(dev, "M", P * T * C)
            pair = need[5] and m == 4 and _GEMM_PAIR and lib.wino4_bwd_gemm_pair_supported(N, H, W, C, K)
            defer = need[5] and m == 4 and _DEFER_DW.wants(w, direct_w)       # dw += G^T dU G joins the step's one deferred launch
            slab_key = ("dUp", w.data_ptr()) if defer else "dUp"
            if pair:
                # 2 + 4a. the backward-data product and the backward-weight product do not depend on each other: ONE launch whose
                # workgroups split between them (hifihr_wino4_bwd_gemm_pair; HIFIHR_GEMM_PAIR=0: two launches)
                parts = lib.wino_wgrad_parts(N, H, W, C, K, m)
                dU = _wino_scratch(dev, slab_key, parts * P * K * C)
                lib.wino4_bwd_gemm_pair(V2, U2, M2, V, Yt, dU, N, H, W, C, K, parts)
            else:
                lib.wino_gemm(V2, U2, M2, N, H, W, K, C, ws=_wino_gemm_ws(lib, dev, N, H, W, K, C, m), m=m)
            # 3. output transform + identity-branch gradient + ReLU mask + batch-norm reduction
            lib.wino_output_transform_bnred(M2, x, res_out, gadd, save_mean, save_invstd, gamma, beta, red, g_in, N, H, W, C, m)
            # 4. backward-weight: dU = Y'^T V over the tiles (slabs, fixed order), dw += G^T dU G
            if need[5]:
                tgt = w.grad if direct_w else torch.zeros_like(wk, memory_format=_CL)
                if not pair:
                    parts = lib.wino_wgrad_parts(N, H, W, C, K, m)
                    dU = _wino_scratch(dev, slab_key, parts * P * K * C)
                    lib.wino_wgrad_gemm_parts(V, Yt, dU, N, H, W, C, K, parts, m)
                if defer and parts > 0:
                    _DEFER_DW.add(dU, parts, tgt, K, C)
                else:
                    lib.wino_dw_transform_parts(dU, parts, tgt, K, C, m)
                dw_box[0] = None if direct_w else tgt
        if PROFILE.on:
            if need[5] and m == 4 and _GEMM_PAIR and lib.wino4_bwd_gemm_pair_supported(N, H, W, C, K):
                PROFILE.conv_log.append((("wino", N, H, W, C, K, m), "gemm-pair"))
            else:
                PROFILE.conv_log.append((("wino", N, H, W, K, C, m), "gemm"))
                PROFILE.conv_log.append((("wino", N, H, W, C, K, m), "gemm-tn"))
        PROFILE.bracket("bn_conv_bwd_wino", run)
        if lazy_in:
            _ZERO_POOL.release(link.red)         # folded and zeroed by the dual transform
            link.red = link.g = None
            link.gamma_param._hifihr_grad_deferred = link.beta_param._hifihr_grad_deferred = False
            _grad_ready(link.gamma_param)
            _grad_ready(link.beta_param)
        if need[5] and direct_w:
            _grad_ready(w)
        # 5. the batch-norm apply: inside the producer's dual transform when x came from a fused function, else one launch here
        dg_ret = db_ret = None
        if in_link is not None and _direct_grad(ctx.gamma_param) and _direct_grad(ctx.beta_param):
            in_link.red, in_link.save_mean, in_link.save_invstd, in_link.gamma = red, save_mean, save_invstd, gamma
            in_link.gamma_param, in_link.beta_param, in_link.g = ctx.gamma_param, ctx.beta_param, g_in
            # (the data-parallel reducer must not take autograd's "this function is done" for "dgamma / dbeta are written": dist.py)
            ctx.gamma_param._hifihr_grad_deferred = ctx.beta_param._hifihr_grad_deferred = True
            dx = g_in                            # NOT d loss / d x yet: see _WinoLink
        else:
            dx = torch.empty_like(x, memory_format=_CL)
            dg_t, dg_ret = acc_target(ctx.gamma_param, C)
            db_t, db_ret = acc_target(ctx.beta_param, C)
            PROFILE.bracket("bn_bwd", lambda: lib.bn_bwd_apply(g_in, x, save_mean, save_invstd, gamma, N * H * W, C, red, dx, dg_t, db_t))
            _ZERO_POOL.release(red)
            if dg_ret is None:
                _grad_ready(ctx.gamma_param)
            if db_ret is None:
                _grad_ready(ctx.beta_param)
        return dx, None, dg_ret, db_ret, (g_in if ctx.has_res else None), dw_box[0], None, None, None, None, None, None, None


def bn_act_wino_conv(x, stats, bn: torch.nn.BatchNorm2d, residual, w, want_stats):
    """See _BNActWinoConv; `bn` in training mode, `stats` from the producer of x (conv2d(..., want_stats=True))."""
    link = _WinoLink()
    y, st, out = _BNActWinoConv.apply(x, stats, bn.weight, bn.bias, residual, w, float(bn.eps), float(bn.momentum), bn.running_mean,
                                      bn.running_var, want_stats, getattr(x, "_hifihr_link", None), link)
    y._hifihr_link = link
    return y, st, out


_ACT = {None: 0, False: 0, True: 1, "relu": 1, "swish": 2}


def bn_act(x, stats, bn: torch.nn.BatchNorm2d, residual=None, relu=True):
    """act(bn(x) + residual?) with train-mode batch statistics; `relu` is True/"relu", "swish", or False/None
    (reference trunks: nn.BatchNorm2d + `out += identity` + nn.ReLU; EfficientNet: BN + MemoryEfficientSwish).
    `stats` comes from conv2d(..., want_stats=True).  Eval mode: the same fused kernel on the running statistics (`stats` must be None).
    bn.num_batches_tracked is not advanced (it only matters for momentum=None, which the reference never uses)."""
    act = _ACT[relu]
    if not bn.training:
        # evaluation mode (reference train_hrnet.py:119-161: model.eval()): the same fused kernel on the running statistics.
        # A producer must not have been asked for batch statistics (callers pass want_stats=bn.training): a slot buffer that
        # nobody consumes would go back to the allocator dirty while a captured hipGraph still holds its address.
        assert stats is None, "bn_act in eval mode: the producer was asked for batch statistics (pass want_stats=bn.training)"
        require_cuda(x)
        lib = get_lib()
        xc = x.contiguous(memory_format=_CL)
        N, C, H, W = xc.shape
        res = residual.contiguous(memory_format=_CL) if residual is not None else None
        return _BNActEval.apply(xc, bn.weight, bn.bias, res, bn.running_mean, bn.running_var, act, float(bn.eps))
    if stats is None:                       # producer was not one of our convolutions: one HBM-bound statistics pass
        require_cuda(x)
        lib = get_lib()
        xc = x.contiguous(memory_format=_CL)
        N, C, H, W = xc.shape
        stats = _ZERO_POOL.acquire(lib.bn_stats_floats(C), x.device)
        PROFILE.bracket("bn_stats", lambda: lib.bn_stats(xc, N * H * W, C, stats))
        x = xc
    return _BNAct.apply(x, stats, bn.weight, bn.bias, residual, act, float(bn.eps), float(bn.momentum),
                        bn.running_mean, bn.running_var)


class _BNReluMaxPool(torch.autograd.Function):
    """MaxPool2d(3, 2, 1)(ReLU(BN(x))) in one forward and two backward launches (csrc/bn.hip: bn_relu_pool_fwd_kernel,
    bn_pool_bwd_reduce_kernel, bn_pool_bwd_apply_kernel): keeps x, the pooled winners' taps and the batch statistics."""

    @staticmethod
    def forward(ctx, x, stats, gamma, beta, eps, momentum, running_mean, running_var):
        require_cuda(x, stats, gamma, beta)
        lib = get_lib()
        x = x.contiguous(memory_format=_CL)
        N, C, H, W = x.shape
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((N, C, OH, OW), device=x.device, memory_format=_CL)
        tap = torch.empty(N * OH * OW * C, dtype=torch.uint8, device=x.device)
        save_mean = torch.empty(C, device=x.device)
        save_invstd = torch.empty(C, device=x.device)
        PROFILE.bracket("bn_pool_fwd", lambda: lib.bn_relu_maxpool_fwd(x, stats, gamma, beta, N, H, W, C, eps, momentum, y, tap, save_mean,
                                                                      save_invstd, running_mean, running_var))
        _ZERO_POOL.release(stats)
        # the pooled output itself goes back into backward: the batch-norm reduction walks IT instead of every input pixel
        # (hifihr_bn_relu_maxpool_bwd_y; HIFIHR_STEM_REDUCE_Y=0: the pass over x).  It is the next convolution's saved input anyway.
        ctx.save_for_backward(x, tap, gamma, beta, save_mean, save_invstd, y if _STEM_REDUCE_Y else None)
        ctx.gamma_param, ctx.beta_param = gamma, beta
        return y

    @staticmethod
    def backward(ctx, gy):
        x, tap, gamma, beta, save_mean, save_invstd, y = ctx.saved_tensors
        lib = get_lib()
        N, C, H, W = x.shape
        gy = gy.contiguous(memory_format=_CL)
        _DEFER_DW.flush_early(x.device)           # behind the stem's pooling only the stem is left: the deferred weight-gradient launch runs beside it
        dx = torch.empty_like(x, memory_format=_CL)
        red = _ZERO_POOL.acquire(lib.bn_stats_floats(C), x.device)
        dg_t, dg_ret = _bn_acc_target(ctx.gamma_param, C, x.device)
        db_t, db_ret = _bn_acc_target(ctx.beta_param, C, x.device)
        if y is not None:
            PROFILE.bracket("bn_pool_bwd", lambda: lib.bn_relu_maxpool_bwd_y(gy, y, tap, x, save_mean, save_invstd, gamma, beta, N, H, W, C, red, dx,
                                                                            dg_t, db_t))
        else:
            PROFILE.bracket("bn_pool_bwd", lambda: lib.bn_relu_maxpool_bwd(gy, tap, x, save_mean, save_invstd, gamma, beta, N, H, W, C, red, dx,
                                                                          dg_t, db_t))
        _ZERO_POOL.release(red)
        if dg_ret is None:
            _grad_ready(ctx.gamma_param)
        if db_ret is None:
            _grad_ready(ctx.beta_param)
        return dx, None, dg_ret, db_ret, None, None, None, None


def _bn_acc_target(p, C, device):
    if getattr(p, "_hifihr_direct_grad", False) and p.grad is not None:
        return p.grad, None                 # accumulate straight into the flat gradient buffer
    t = torch.zeros(C, device=device)
    return t, t


def bn_relu_maxpool(x, stats, bn: torch.nn.BatchNorm2d):
    """nn.MaxPool2d(3, 2, 1)(relu(bn(x))): the ResNet stem behind conv1 (reference trunk: vendored resnet.py forward).  Training
    mode with batch statistics from our convolution and C <= 512: the fused kernels; anything else: bn_act followed by maxpool3x3s2.
    HIFIHR_BN_POOL=0 keeps the two-step path (A/B timing)."""
    if bn.training and stats is not None and x.is_cuda and os.environ.get("HIFIHR_BN_POOL", "1") != "0":
        N, C, H, W = x.shape
        if get_lib().bn_relu_maxpool_supported(N, H, W, C):
            return _BNReluMaxPool.apply(x, stats, bn.weight, bn.bias, float(bn.eps), float(bn.momentum), bn.running_mean, bn.running_var)
    return maxpool3x3s2(bn_act(x, stats, bn, None, True))


# ------------------------------------------------------------------------------------------------
# fused losses (csrc/losses.hip)
# ------------------------------------------------------------------------------------------------
GEOM_TERMS = ("joint_3d", "vert_3d", "edge_length", "mshape", "mpose")
_VF_CACHE = {}


def _vertex_face_csr(faces, V):
    """faces [F,3] int32 (device) -> (vf_off [V+1], vf_idx [3F]) int32 on the same device: for every vertex the incident
    (face * 4 + corner) entries in ascending face order (same table as the renderer builds, hifihr_api.hip)."""
    key = (faces.data_ptr(), int(faces.shape[0]), V, str(faces.device))
    hit = _VF_CACHE.get(key)
    if hit is None:
        import numpy as np
        f = faces.detach().cpu().numpy().astype(np.int64)
        flat = f.reshape(-1)
        order = np.argsort(flat, kind="stable")                      # stable: ascending face order within a vertex
        off = np.zeros(V + 1, dtype=np.int32)
        np.add.at(off, flat + 1, 1)
        off = np.cumsum(off).astype(np.int32)
        idx = ((order // 3) * 4 + (order % 3)).astype(np.int32)
        hit = (torch.from_numpy(off).to(faces.device), torch.from_numpy(idx).to(faces.device))
        _VF_CACHE[key] = hit
    return hit


class _GeomLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, joints, joints_gt, verts, verts_gt, shape, pose, faces, mse, lam):
        require_cuda(joints, verts)
        lib = get_lib()
        joints, joints_gt, verts, verts_gt = (t.contiguous().float() for t in (joints, joints_gt, verts, verts_gt))
        shape = shape.contiguous() if shape is not None else None
        pose = pose.contiguous() if pose is not None else None
        B = joints.shape[0]
        partial = torch.empty(B * 5, device=joints.device)
        out = torch.empty(5, device=joints.device)
        PROFILE.bracket("geom_loss_fwd", lambda: lib.geom_loss_fwd(joints, joints_gt, verts, verts_gt, shape, pose, faces, mse, lam,
                                                                  partial, out))
        ctx.save_for_backward(joints, joints_gt, verts, verts_gt, shape, pose, faces)
        ctx.mse, ctx.lam = mse, tuple(lam)
        return out

    @staticmethod
    def backward(ctx, gout):
        joints, joints_gt, verts, verts_gt, shape, pose, faces = ctx.saved_tensors
        lib = get_lib()
        need = ctx.needs_input_grad
        gj = torch.empty_like(joints) if need[0] else None
        gv = torch.empty_like(verts) if need[2] else None
        gs = torch.empty_like(shape) if (shape is not None and need[4]) else None
        gp = torch.empty_like(pose) if (pose is not None and need[5]) else None
        vf_off, vf_idx = _vertex_face_csr(faces, verts.shape[1]) if faces is not None else (None, None)
        gout = gout.contiguous()
        PROFILE.bracket("geom_loss_bwd", lambda: lib.geom_loss_bwd(joints, joints_gt, verts, verts_gt, shape, pose, faces, vf_off, vf_idx,
                                                                  ctx.mse, ctx.lam, gout, gj, gv, gs, gp))
        return gj, None, gv, None, gs, gp, None, None, None


def geom_losses(joints, joints_gt, verts, verts_gt, shape, pose, faces, mse, lam):
    """[5] = lambda-weighted (joint_3d, vert_3d, edge_length, mshape, mpose) of reference losses.py:259-266, 283-284,
    398-406 in one launch (+ finisher); `faces` [F,3] int32 or None, `lam` the five lambdas (0 for unused terms)."""
    return _GeomLoss.apply(joints, joints_gt, verts, verts_gt, shape, pose, faces, bool(mse), tuple(float(v) for v in lam))


class _LossTotal(torch.autograd.Function):
    """sum of the first counts[i] entries of each part (the fused loss kernels' output vectors / scalars): one launch each way
    (hifihr_loss_total_fwd / _bwd) instead of stack + sum and, in backward, a cat per vector plus a zero fill."""

    @staticmethod
    def forward(ctx, counts, *parts):
        require_cuda(*parts)
        ctx.orig = [p.shape for p in parts]
        parts = [p.contiguous().reshape(-1) for p in parts]
        total = torch.empty((), device=parts[0].device)
        PROFILE.bracket("loss_total", lambda: get_lib().loss_total_fwd(parts, counts, total))
        ctx.counts, ctx.shapes = counts, [p.shape for p in parts]
        return total

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        grads = [torch.empty(s, device=g.device) for s in ctx.shapes]
        PROFILE.bracket("loss_total", lambda: get_lib().loss_total_bwd(g, grads, ctx.counts))
        return (None, *[gr.reshape(s) for gr, s in zip(grads, ctx.orig)])


def loss_total(parts):
    """parts: [(vector or 0-d tensor, how many of its leading entries are loss terms)] -> their sum as a 0-d tensor."""
    return _LossTotal.apply(tuple(int(n) for _, n in parts), *[p for p, _ in parts])


class _JointTerms(torch.autograd.Function):
    @staticmethod
    def forward(ctx, j2d, j2d_gt, joints, joints_gt, mse, lam3):
        lib = get_lib()
        prep = lambda t: t.contiguous().float() if t is not None else None
        j2d, j2d_gt, joints, joints_gt = prep(j2d), prep(j2d_gt), prep(joints), prep(joints_gt)
        require_cuda(*(t for t in (j2d, joints) if t is not None))
        ref = j2d if j2d is not None else joints
        out = torch.empty(3, device=ref.device)
        PROFILE.bracket("joint_terms_fwd", lambda: lib.joint_terms_fwd(j2d, j2d_gt, joints, joints_gt, mse, lam3, out))
        ctx.save_for_backward(*(t if t is not None else ref.new_empty(0) for t in (j2d, j2d_gt, joints, joints_gt)))
        ctx.has2, ctx.has3, ctx.mse, ctx.lam3 = j2d is not None, joints is not None, mse, lam3
        return out

    @staticmethod
    def backward(ctx, gout):
        j2d, j2d_gt, joints, joints_gt = (t if t.numel() else None for t in ctx.saved_tensors)
        lib = get_lib()
        g2 = torch.empty_like(j2d) if (ctx.has2 and ctx.needs_input_grad[0]) else None
        g3 = torch.empty_like(joints) if (ctx.has3 and ctx.needs_input_grad[2]) else None
        if g2 is not None or g3 is not None:
            PROFILE.bracket("joint_terms_bwd", lambda: lib.joint_terms_bwd(j2d, j2d_gt, joints, joints_gt, ctx.mse, ctx.lam3, gout.contiguous(), g2, g3))
        return g2, None, g3, None, None, None


def joint_terms(j2d, j2d_gt, joints, joints_gt, mse, lam3):
    """[3] = lambda-weighted (joint_2d, bone_direc, bone_direc_3d) of reference losses.py:267-282 in one launch (csrc/losses.hip);
    (j2d, j2d_gt) [B,21,2] or (None, None), (joints, joints_gt) [B,21,3] or (None, None); lam3 = the three lambdas (0 = unused)."""
    return _JointTerms.apply(j2d, j2d_gt, joints, joints_gt, bool(mse), tuple(float(v) for v in lam3))


class _PhotoLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgba, imgs, seg, l_tex, l_mrgb, l_sil):
        require_cuda(rgba, imgs, seg)
        lib = get_lib()
        rgba, imgs, seg = rgba.contiguous(), imgs.contiguous(), seg.contiguous()
        B, _, H, W = rgba.shape
        re_img_m = torch.empty(B, 3, H, W, device=rgba.device)
        mask_rgbs = torch.empty_like(re_img_m)
        partial = torch.empty(lib.photo_loss_partial_floats(), device=rgba.device)
        out = torch.empty(4, device=rgba.device)
        PROFILE.bracket("photo_loss_fwd", lambda: lib.photo_loss_fwd(rgba, imgs, seg, l_tex, l_mrgb, l_sil, re_img_m, mask_rgbs, partial,
                                                                    out))
        ctx.save_for_backward(rgba, re_img_m, mask_rgbs, out)
        ctx.lams = (l_tex, l_mrgb)
        ctx.mark_non_differentiable(mask_rgbs)
        ctx.set_materialize_grads(False)
        return out, re_img_m, mask_rgbs

    @staticmethod
    def backward(ctx, gout, g_re_img, _gm):
        if gout is None and g_re_img is None:
            return (None,) * 6
        rgba, re_img_m, mask_rgbs, out = ctx.saved_tensors
        grad = torch.empty_like(rgba)
        gout = gout.contiguous() if gout is not None else None
        g_re_img = g_re_img.contiguous() if g_re_img is not None else None
        PROFILE.bracket("photo_loss_bwd", lambda: get_lib().photo_loss_bwd(rgba, re_img_m, mask_rgbs, g_re_img, gout, out, ctx.lams[0],
                                                                          ctx.lams[1], grad))
        return grad, None, None, None, None, None


def photo_losses(rgba, imgs, seg, l_tex, l_mrgb, l_sil):
    """The photometric block of reference losses.py:355-378 (+ the `sil` term, :388-390) from the renderer's rgba:
    returns (out[4] = lambda-weighted texture, mrgb, sil and mean(re_img) - mean(mask_rgbs); re_img (masked, feeds SSIM);
    mask_rgbs)."""
    return _PhotoLoss.apply(rgba, imgs, seg, float(l_tex), float(l_mrgb), float(l_sil))


def sil_post(rgba, images):
    """re_sil = where(alpha > 0, 255, alpha) [B,1,H,W] and maskRGBs = images * (re_sil > 0) (models_res_nimble.py:219-220);
    no gradient (the reference detaches the silhouette)."""
    require_cuda(rgba, images)
    rgba = rgba.detach().contiguous()
    B, _, H, W = rgba.shape
    re_sil = torch.empty(B, 1, H, W, device=rgba.device)
    mask_rgbs = torch.empty(B, 3, H, W, device=rgba.device)
    get_lib().sil_post(rgba, images.contiguous(), re_sil, mask_rgbs)
    return re_sil, mask_rgbs


# ------------------------------------------------------------------------------------------------
# small-batch fully connected layers of the heads (csrc/mlp.hip)
# ------------------------------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, act, eps, momentum, running_mean, running_var):
        require_cuda(x, w)
        lib = get_lib()
        x, w = x.contiguous(), w.contiguous()
        B, O = x.shape[0], w.shape[0]
        y = torch.empty(B, O, device=x.device)
        bn = None
        z = sm = si = None
        if gamma is not None:
            z, sm, si = torch.empty(B, O, device=x.device), torch.empty(O, device=x.device), torch.empty(O, device=x.device)
            bn = (gamma, beta, eps, momentum, running_mean, running_var, z, sm, si)
        PROFILE.bracket("linear_fwd", lambda: lib.linear_fwd(x, w, b, act, y, bn))
        ctx.save_for_backward(x, w, y if act == 1 else None, gamma, z, sm, si)
        ctx.act = act
        ctx.params = (w, b, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y, gamma, z, sm, si = ctx.saved_tensors
        pw, pb, pg, pbeta = ctx.params
        lib = get_lib()
        dy = dy.contiguous()
        B, O = dy.shape
        need_dx = ctx.needs_input_grad[0]
        dz = torch.empty(B, O, device=dy.device)
        dx = torch.empty_like(x) if need_dx else None
        dw_t, dw_ret = _acc_target(pw, pw.shape, dy.device)
        db_t, db_ret = _acc_target(pb, pb.shape, dy.device) if pb is not None else (None, None)
        bn = None
        dg_ret = dbt_ret = None
        if gamma is not None:
            dg_t, dg_ret = _acc_target(pg, pg.shape, dy.device)
            dbt_t, dbt_ret = _acc_target(pbeta, pbeta.shape, dy.device)
            bn = (gamma, z, sm, si, dg_t, dbt_t)
        PROFILE.bracket("linear_bwd", lambda: lib.linear_bwd(dy, y, x, w, ctx.act, dz, dw_t, db_t, dx, bn))
        for p, ret in ((pw, dw_ret), (pb, db_ret), (pg, dg_ret), (pbeta, dbt_ret)):
            if p is not None and ret is None:
                _grad_ready(p)
        return dx, dw_ret, db_ret, dg_ret, dbt_ret, None, None, None, None, None


class _LinearGroup(torch.autograd.Function):
    """n independent act(x_i W_i^T + b_i) in ONE launch forward and TWO backward (csrc/mlp.hip grouped kernels)."""

    @staticmethod
    def forward(ctx, n, acts, *tensors):
        xs, ws, bs = tensors[:n], tensors[n:2 * n], tensors[2 * n:3 * n]
        require_cuda(*xs, *ws)
        lib = get_lib()
        xs = [x.contiguous() for x in xs]
        ws_c = [w.contiguous() for w in ws]
        ys = [torch.empty(x.shape[0], w.shape[0], device=x.device) for x, w in zip(xs, ws_c)]
        members = [dict(x=x, w=w, b=b, y=y, act=a) for x, w, b, y, a in zip(xs, ws_c, bs, ys, acts)]
        PROFILE.bracket("linear_fwd", lambda: lib.linear_fwd_group(members))
        ctx.n, ctx.acts, ctx.params = n, acts, (ws, bs)
        ctx.save_for_backward(*xs, *ws_c, *ys)
        ctx.set_materialize_grads(False)           # heads whose output no loss reads get no backward work at all
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        n = ctx.n
        saved = ctx.saved_tensors
        xs, ws_c, ys = saved[:n], saved[n:2 * n], saved[2 * n:3 * n]
        pws, pbs = ctx.params
        lib = get_lib()
        dxs, dws, dbs = [None] * n, [None] * n, [None] * n
        members, ready = [], []
        shared = {}                                    # input storage -> the ONE dx its members add into
        for i in range(n):
            if dys[i] is None:
                continue
            dy = dys[i].contiguous()
            dx = None
            if ctx.needs_input_grad[2 + i]:
                # members that read the SAME input (the heads' first layers all read the base features) share one dx: the backward-weight
                # kernel zero-fills it, the backward-data kernel adds every member's product into it with atomics -- autograd would
                # otherwise sum the members' gradients with an elementwise launch per extra member
                key = (xs[i].data_ptr(), tuple(xs[i].shape))
                dx = shared.get(key)
                first = dx is None
                if first:
                    dx = shared[key] = torch.empty_like(xs[i])
            dw_t, dws[i] = _acc_target(pws[i], pws[i].shape, dy.device)
            db_t = None
            if pbs[i] is not None:
                db_t, dbs[i] = _acc_target(pbs[i], pbs[i].shape, dy.device)
            dxs[i] = dx if (dx is not None and first) else None          # (returned once: the shared buffer holds the sum)
            members.append(dict(x=xs[i], w=ws_c[i], y=ys[i], act=ctx.acts[i], dy=dy, dz=torch.empty_like(dy), dW=dw_t, db=db_t, dx=dx))
            ready += [p for p, ret in ((pws[i], dws[i]), (pbs[i], dbs[i])) if p is not None and ret is None]
        if members:
            PROFILE.bracket("linear_bwd", lambda: lib.linear_bwd_group(members))
        for p in ready:
            _grad_ready(p)
        return (None, None, *dxs, *dws, *dbs)


def linear_group(members):
    """[(x, nn.Linear, relu?)] -> [act(lin(x))]: up to six independent layers of the same depth as one launch (the regression
    heads of the HandEncoder, reference network/res_encoder.py:112-131)."""
    n = len(members)
    acts = tuple(1 if m[2] in (True, "relu", 1) else 0 for m in members)
    return list(_LinearGroup.apply(n, acts, *[m[0] for m in members], *[m[1].weight for m in members], *[m[1].bias for m in members]))


class _TexturePCA(torch.autograd.Function):
    @staticmethod
    def forward(ctx, coef, basis, mean):
        require_cuda(coef, basis)
        coef, basis = coef.contiguous(), basis.contiguous()
        out = torch.empty(coef.shape[0], basis.shape[1], device=coef.device)
        PROFILE.bracket("texture_pca_fwd", lambda: get_lib().texture_pca_fwd(coef, basis, mean, out))
        ctx.save_for_backward(basis)
        ctx.K = coef.shape[1]
        return out

    @staticmethod
    def backward(ctx, g):
        basis, = ctx.saved_tensors
        g = g.contiguous()
        dcoef = torch.zeros(g.shape[0], ctx.K, device=g.device)
        PROFILE.bracket("texture_pca_bwd", lambda: get_lib().texture_pca_bwd(g, basis, dcoef))
        return dcoef, None, None


def texture_pca_decode(coef, basis, mean=None):
    """mean + coef . basis: coef [B,K], basis [K,n] (constant), mean [n] -> [B,n]; n % 4 == 0.  The NIMBLE texture decode (reference
    models_res_nimble.py:133-142 consumes its result), one HBM-bound launch per direction (csrc/texpca.hip)."""
    return _TexturePCA.apply(coef, basis, mean)


class _LightSplit(torch.autograd.Function):
    """lights [B, 6] -> (hardtanh(lights[:, :3]), lights[:, 3:]) as two CONTIGUOUS tensors (reference network/res_encoder.py:205-210:
    `colors = self.hardtanh(lights[:, :3]); directions = lights[:, 3:]`): one launch each way (hifihr_light_split_fwd / _bwd; round 4:
    clamp + copy forward, hardtanh_backward + cat backward; as separate autograd nodes seven launches in the backward)."""

    @staticmethod
    def forward(ctx, lights):
        require_cuda(lights)
        lights = lights.contiguous()
        B = lights.shape[0]
        colors, directions = torch.empty(B, 3, device=lights.device), torch.empty(B, 3, device=lights.device)
        get_lib().light_split_fwd(lights, colors, directions)
        ctx.save_for_backward(lights)
        ctx.set_materialize_grads(False)
        return colors, directions

    @staticmethod
    def backward(ctx, gc, gd):
        lights, = ctx.saved_tensors
        gl = torch.empty_like(lights)
        c = lambda t: t.contiguous() if t is not None else None
        get_lib().light_split_bwd(lights, c(gc), c(gd), gl)
        return gl


def light_split(lights):
    return _LightSplit.apply(lights)


def affine(x, w, b):
    """x[B, I] . w[O, I]^T + b[O] with constant (buffer) w and b, one launch of the small-batch linear kernel."""
    return _Linear.apply(x, w, b, None, None, 0, 0.0, 0.0, None, None)


def linear(x, lin: torch.nn.Linear, act=None, bn: torch.nn.BatchNorm1d | None = None):
    """act(bn(lin(x))) for a [B, I] activation in ONE launch (two backward): nn.Linear (+ nn.BatchNorm1d, batch statistics
    in training mode) (+ nn.ReLU) of the reference's regression heads.  The fused batch-norm epilogue needs every row of the batch
    in one workgroup (B <= 64); larger batches and evaluation mode run the layer as linear -> bn_act (csrc/bn.hip on a [B, O, 1, 1]
    view: the same train- / eval-mode batch-norm kernels the trunk uses), still hand-written kernels only."""
    a = 1 if act in (True, "relu", 1) else 0
    if bn is not None and (not bn.training or x.shape[0] > 64):
        z = _Linear.apply(x, lin.weight, lin.bias, None, None, 0, 0.0, 0.0, None, None)
        B, O = z.shape
        return bn_act(z.view(B, O, 1, 1), None, bn, None, bool(a)).reshape(B, O)
    if bn is None:
        return _Linear.apply(x, lin.weight, lin.bias, None, None, a, 0.0, 0.0, None, None)
    return _Linear.apply(x, lin.weight, lin.bias, bn.weight, bn.bias, a, float(bn.eps), float(bn.momentum), bn.running_mean,
                         bn.running_var)


# ------------------------------------------------------------------------------------------------
# squeeze-and-excitation (csrc/se.hip + the small-batch linear kernels)
# ------------------------------------------------------------------------------------------------
def _se_w2t(w2):
    """The expand weight W2[C][SQ] transposed to [SQ][C] (the fused kernels read it coalesced over the channels): from the per-step
    re-layout launch inside `prepared_weights()`, by one ATen transpose outside it (evaluation, the first step)."""
    C, SQ = w2.shape[0], w2.shape[1]
    t = _WEIGHT_PREP.get(w2, w2, 0) if w2.dim() == 4 else None
    return t if t is not None else w2.reshape(C, SQ).t().contiguous()


class _SqueezeExcite(torch.autograd.Function):
    """Pooling -> the two layers in ONE launch (csrc/se.hip se_mlp_fwd_kernel) -> scaling: 3 launches forward, 4 backward, no fills (round
    3: 5 + 7 with the layers on the head kernels; HIFIHR_SE_FUSED=0 keeps that form for the A/B)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        require_cuda(x, w1, w2)
        lib = get_lib()
        x = x.contiguous(memory_format=_CL)
        B, C, H, W = x.shape
        HW, SQ = H * W, w1.shape[0]
        dev = x.device
        fused = _SE_FUSED and lib.se_mlp_supported(C, SQ)
        ctx.fused = fused
        h1, z1, gate = torch.empty(B, SQ, device=dev), torch.empty(B, SQ, device=dev), torch.empty(B, C, device=dev)
        if fused:
            acc = _ZERO_POOL.acquire(B * C, dev)                  # handed back zeroed by se_mlp_fwd
            PROFILE.bracket("se_pool", lambda: lib.se_pool(x, B, HW, C, acc))
            mean = torch.empty(B, C, device=dev)
            w2t = _se_w2t(w2)
            PROFILE.bracket("se_mlp_fwd", lambda: lib.se_mlp_fwd(acc, w1, b1, w2t, b2, B, C, SQ, mean, z1, h1, gate))
            _ZERO_POOL.release(acc)
        else:
            mean = torch.zeros(B, C, device=dev)
            PROFILE.bracket("se_pool", lambda: lib.se_pool(x, B, HW, C, mean))
            PROFILE.bracket("linear_fwd", lambda: lib.linear_fwd(mean, w1, b1, 2, h1, z=z1))          # swish
            PROFILE.bracket("linear_fwd", lambda: lib.linear_fwd(h1, w2, b2, 3, gate))                 # sigmoid
        y = torch.empty_like(x, memory_format=_CL)
        PROFILE.bracket("se_scale", lambda: lib.se_scale(x, gate, None, 0.0, B, HW, C, y))
        ctx.save_for_backward(x, mean, h1, z1, gate, w1, w2)
        ctx.params = (w1, b1, w2, b2)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, h1, z1, gate, w1, w2 = ctx.saved_tensors
        pw1, pb1, pw2, pb2 = ctx.params
        lib = get_lib()
        dy = dy.contiguous(memory_format=_CL)
        B, C, H, W = x.shape
        HW, SQ = H * W, w1.shape[0]
        dev = x.device
        rets = []
        tgts = []
        for p in (pw1, pb1, pw2, pb2):
            t, r = _acc_target(p, p.shape, dev)
            tgts.append(t); rets.append(r)
        dmean = torch.empty(B, C, device=dev)
        dz2, dz1 = torch.empty(B, C, device=dev), torch.empty(B, SQ, device=dev)
        if ctx.fused:
            dgate = _ZERO_POOL.acquire(B * C, dev)                # handed back zeroed by se_mlp_bwd
            PROFILE.bracket("se_bwd_gate", lambda: lib.se_bwd_gate(dy, x, B, HW, C, dgate))
            w2t = _se_w2t(pw2)
            PROFILE.bracket("se_mlp_bwd", lambda: lib.se_mlp_bwd(dgate, gate, z1, h1, mean, w1, w2t, B, C, SQ, dz2, dz1, dmean, tgts[0], tgts[1],
                                                                 tgts[2], tgts[3]))
            _ZERO_POOL.release(dgate)
        else:
            dgate = torch.zeros(B, C, device=dev)
            PROFILE.bracket("se_bwd_gate", lambda: lib.se_bwd_gate(dy, x, B, HW, C, dgate))
            dh1 = torch.empty(B, SQ, device=dev)
            PROFILE.bracket("linear_bwd", lambda: lib.linear_bwd(dgate, gate, h1, w2, 3, dz2, tgts[2], tgts[3], dh1))
            PROFILE.bracket("linear_bwd", lambda: lib.linear_bwd(dh1, None, mean, w1, 2, dz1, tgts[0], tgts[1], dmean, z=z1))
        dx = torch.empty_like(x, memory_format=_CL)
        PROFILE.bracket("se_bwd_dx", lambda: lib.se_scale(dy, gate, dmean, 1.0 / HW, B, HW, C, dx))
        for p, r in zip((pw1, pb1, pw2, pb2), rets):
            if r is None:
                _grad_ready(p)
        return dx, rets[0], rets[1], rets[2], rets[3]


_SE_FUSED = os.environ.get("HIFIHR_SE_FUSED", "1") != "0"


class _DropConnectAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, skip, u, keep):
        require_cuda(x, skip, u)
        x, skip = x.contiguous(memory_format=_CL), skip.contiguous(memory_format=_CL)
        u = u.reshape(-1).contiguous().float()
        B = x.shape[0]
        out = torch.empty_like(x, memory_format=_CL)
        PROFILE.bracket("drop_connect_add", lambda: get_lib().drop_connect_add(x, skip, u, keep, B, x.numel() // B, out))
        ctx.save_for_backward(u)
        ctx.keep = keep
        return out

    @staticmethod
    def backward(ctx, dy):
        (u,) = ctx.saved_tensors
        dy = dy.contiguous(memory_format=_CL)
        B = dy.shape[0]
        dx = torch.empty_like(dy, memory_format=_CL)
        PROFILE.bracket("drop_connect_add", lambda: get_lib().drop_connect_add(dy, None, u, ctx.keep, B, dy.numel() // B, dx))
        return dx, dy, None, None


def drop_connect_add(x, skip, u, keep):
    """x / keep * floor(keep + u[b]) + skip: the drop-connect and the skip connection of an MBConv block (reference
    network/efficientnet_pt/utils.py:82-91, model.py:91-94) as one launch per direction (u: the block's per-sample uniform draws)."""
    if x.numel() // x.shape[0] % 4 != 0:
        return x / keep * torch.floor(keep + u.reshape(-1, 1, 1, 1)) + skip
    return _DropConnectAdd.apply(x, skip, u, float(keep))


def squeeze_excite(x, reduce_conv, expand_conv):
    """x * sigmoid(expand(swish(reduce(mean_hw(x))))) for channels_last x; reduce / expand are the block's 1x1 nn.Conv2d with
    bias (reference network/efficientnet_pt/model.py:82-86): 3 launches forward, 4 backward."""
    return _SqueezeExcite.apply(x, reduce_conv.weight, reduce_conv.bias, expand_conv.weight, expand_conv.bias)


# ------------------------------------------------------------------------------------------------
# pooling (csrc/pool.hip)
# ------------------------------------------------------------------------------------------------
def _acc_target(p, shape, device):
    """(tensor the kernel accumulates into, tensor to hand back to autograd or None): straight into the flat gradient
    buffer when the parameter lives in one (hifihr_amd/optim.py FlatParams)."""
    if getattr(p, "_hifihr_direct_grad", False) and p.grad is not None:
        return p.grad, None
    t = torch.zeros(shape, device=device)
    return t, t


class _MMPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p):
        require_cuda(x, p)
        lib = get_lib()
        x = x.contiguous(memory_format=_CL)
        B, C, H, W = x.shape
        y = torch.empty(B, C, device=x.device)
        argmax = torch.empty(B, C, dtype=torch.int32, device=x.device)
        xmax, xavg = torch.empty_like(y), torch.empty_like(y)
        PROFILE.bracket("mmpool_fwd", lambda: lib.mmpool_fwd(x, p, B, H * W, C, y, argmax, xmax, xavg))
        ctx.save_for_backward(p, argmax, xmax, xavg)
        ctx.shape, ctx.p_param = (B, C, H, W), p
        return y

    @staticmethod
    def backward(ctx, gy):
        p, argmax, xmax, xavg = ctx.saved_tensors
        B, C, H, W = ctx.shape
        gy = gy.contiguous()
        dx = torch.empty((B, C, H, W), device=gy.device, memory_format=_CL) if ctx.needs_input_grad[0] else None
        dp_t, dp_ret = _acc_target(ctx.p_param, p.shape, gy.device) if ctx.needs_input_grad[1] else (None, None)
        if dx is None:
            raise NotImplementedError("mmpool backward without an input gradient")
        PROFILE.bracket("mmpool_bwd", lambda: get_lib().mmpool_bwd(gy, p, argmax, xmax, xavg, B, H * W, C, dx, dp_t))
        if dp_t is not None and dp_ret is None:
            _grad_ready(ctx.p_param)
        return dx, dp_ret


def mmpool(x, p):
    """MMPool((1,1)) (reference network/res_encoder.py:247-265): [B,C,H,W] channels_last -> [B,C]."""
    return _MMPool.apply(x, p)


class _MaxPool2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, s, p, relu_input=False):
        require_cuda(x)
        x = x.contiguous(memory_format=_CL)
        N, C, H, W = x.shape
        OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        y = torch.empty((N, C, OH, OW), device=x.device, memory_format=_CL)
        tap = torch.empty(N * OH * OW * C, dtype=torch.uint8, device=x.device)
        PROFILE.bracket("maxpool_fwd", lambda: get_lib().maxpool2d_fwd(x, N, H, W, C, k, s, p, y, tap))
        ctx.save_for_backward(tap, y if relu_input else None)
        ctx.cfg = (N, C, H, W, k, s, p)
        return y

    @staticmethod
    def backward(ctx, gy):
        tap, y = ctx.saved_tensors
        N, C, H, W, k, s, p = ctx.cfg
        gy = gy.contiguous(memory_format=_CL)
        dx = torch.empty((N, C, H, W), device=gy.device, memory_format=_CL)
        PROFILE.bracket("maxpool_bwd", lambda: get_lib().maxpool2d_bwd(gy, tap, N, H, W, C, k, s, p, dx, relu_y=y))
        return dx, None, None, None, None


def maxpool2d(x, k, s, p, relu_input=False):
    """nn.MaxPool2d(k, s, p) on channels_last activations, (k, s, p) in {(3,2,1), (3,1,1), (2,2,0)}.  relu_input: x is a ReLU's output and
    the gradient returned is ALSO multiplied by the ReLU's [x > 0] (only the winning taps receive anything, and there x = the pooled
    value): the producer of x is then told `grad_premasked`."""
    return _MaxPool2d.apply(x, k, s, p, relu_input)


class _MaxPool2dFlat(torch.autograd.Function):
    """nn.MaxPool2d(k, s, p)(x).view(N, -1): the pooled tensor leaves as the NCHW-ordered matrix the reference's flatten makes (and the
    gradient arrives as one) -- a reshape of the channels-last tensor is a copy kernel in each direction (hifihr_maxpool2d_fwd_flat)."""

    @staticmethod
    def forward(ctx, x, k, s, p):
        require_cuda(x)
        x = x.contiguous(memory_format=_CL)
        N, C, H, W = x.shape
        OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        y = torch.empty((N, C * OH * OW), device=x.device)
        tap = torch.empty(N * OH * OW * C, dtype=torch.uint8, device=x.device)
        PROFILE.bracket("maxpool_fwd", lambda: get_lib().maxpool2d_fwd_flat(x, N, H, W, C, k, s, p, y, tap))
        ctx.save_for_backward(tap)
        ctx.cfg = (N, C, H, W, k, s, p)
        return y

    @staticmethod
    def backward(ctx, gy):
        (tap,) = ctx.saved_tensors
        N, C, H, W, k, s, p = ctx.cfg
        gy = gy.contiguous()
        dx = torch.empty((N, C, H, W), device=gy.device, memory_format=_CL)
        PROFILE.bracket("maxpool_bwd", lambda: get_lib().maxpool2d_bwd_flat(gy, tap, N, H, W, C, k, s, p, dx))
        return dx, None, None, None


def maxpool2d_flatten(x, k, s, p):
    """nn.MaxPool2d(k, s, p)(x).view(x.shape[0], -1) without the reshape's copy (see _MaxPool2dFlat); C % 4 == 0."""
    return _MaxPool2dFlat.apply(x, k, s, p)


def maxpool3x3s2(x):
    """nn.MaxPool2d(3, 2, 1) on channels_last activations (the ResNet stem pool)."""
    return _MaxPool2d.apply(x, 3, 2, 1)


def image_to_nhwc4(images, pad4=None, normalize=True):
    """normalize_batch_3C + repack: [B,3,H,W] -> logical [B,4,H,W] channels_last (4th channel zero).  pad4 = (left, right, top,
    bottom) adds an explicit zero border (EfficientNet stem: static "same" padding), normalize=False skips the normalisation."""
    require_cuda(images)
    B, _, H, W = images.shape
    if pad4 is None and normalize:
        out = torch.empty((B, 4, H, W), device=images.device, dtype=torch.float32, memory_format=_CL)
        get_lib().image_to_nhwc4(images.contiguous(), out)
        return out
    pl, pr, pt, pb = pad4 if pad4 is not None else (0, 0, 0, 0)
    out = torch.empty((B, 4, H + pt + pb, W + pl + pr), device=images.device, dtype=torch.float32, memory_format=_CL)
    get_lib().image_to_nhwc4_padded(images.contiguous(), out, (pl, pr, pt, pb), normalize)
    return out


# ------------------------------------------------------------------------------------------------
# fused SSIM
# ------------------------------------------------------------------------------------------------
def _ssim_window():
    """The 1-D window exactly as pytorch_ssim.gaussian(11, 1.5) builds it (float32 tensor ops), as a ctypes array."""
    import ctypes
    import math
    g = torch.Tensor([math.exp(-(x - 11 // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(11)])
    g = g / g.sum()
    return (ctypes.c_float * 11)(*[float(v) for v in g])


_SSIM_WIN = None


class _SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2, loss_lambda):
        """loss_lambda None: the SSIM value; a float: lambda * (1 - SSIM) (the ssim_tex term) with the scalar glue folded
        into two launches forward and one backward instead of eight."""
        global _SSIM_WIN
        require_cuda(img1, img2)
        lib = get_lib()
        if _SSIM_WIN is None:
            _SSIM_WIN = _ssim_window()
        img1, img2 = img1.contiguous(), img2.contiguous()
        B, C, H, W = img1.shape
        need = ctx.needs_input_grad[0]
        partial = torch.empty(lib.ssim_partial_count(B * C, H, W), device=img1.device)
        maps = torch.empty(3, B, C, H, W, device=img1.device) if need else None
        PROFILE.bracket("ssim_fwd", lambda: lib.ssim_fwd(_SSIM_WIN, img1, img2, partial, maps[0] if need else None,
                                                          maps[1] if need else None, maps[2] if need else None))
        if need:
            ctx.save_for_backward(img1, img2, maps)
        n = float(B * C * H * W)
        out = torch.empty((), device=img1.device)
        if loss_lambda is None:
            ctx.out_scale = 1.0
            lib.ssim_finish(partial, 1.0 / n, 0.0, out)                    # SSIM = sum / n
        else:
            ctx.out_scale = -loss_lambda
            lib.ssim_finish(partial, -loss_lambda / n, loss_lambda, out)   # lambda - lambda * sum / n
        return out

    @staticmethod
    def backward(ctx, g):
        img1, img2, maps = ctx.saved_tensors
        gimg1 = torch.empty_like(img1)
        # the kernel multiplies the incoming gradient by out_scale and divides by the element count itself
        gs = g.reshape(1) if (g.dtype == torch.float32 and g.is_contiguous()) else g.reshape(1).float().contiguous()
        PROFILE.bracket("ssim_bwd", lambda: get_lib().ssim_bwd_scaled(_SSIM_WIN, img1, img2, maps[0], maps[1], maps[2], gs, ctx.out_scale,
                                                                       gimg1))
        return gimg1, None, None


def ssim(img1, img2):
    """pytorch_ssim.ssim(img1, img2) (reference utils/pytorch_ssim/__init__.py:65-73); gradient flows to img1 only
    (img2 is ground-truth data at the reference's call site, losses.py:375)."""
    if img2.requires_grad:
        raise NotImplementedError("fused SSIM differentiates with respect to img1 only")
    return _SSIM.apply(img1, img2, None)


def ssim_loss(img1, img2, lam):
    """lam * (1 - ssim(img1, img2)): the `ssim_tex` term of reference losses.py:375-377 with the scalar arithmetic folded in."""
    if img2.requires_grad:
        raise NotImplementedError("fused SSIM differentiates with respect to img1 only")
    return _SSIM.apply(img1, img2, float(lam))


# ------------------------------------------------------------------------------------------------
# depthwise convolution (EfficientNet MBConv), channels_last activations, weight [C,1,K,K]
# ------------------------------------------------------------------------------------------------
class _DwConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, pad4, want_stats=False):
        require_cuda(x, w)
        lib = get_lib()
        x = x.contiguous(memory_format=_CL)
        w = w.contiguous()
        N, C, H, W = x.shape
        K = w.shape[-1]
        pl, pr, pt, pb = pad4
        OH, OW = (H + pt + pb - K) // stride + 1, (W + pl + pr - K) // stride + 1
        y = torch.empty((N, C, OH, OW), device=x.device, dtype=torch.float32, memory_format=_CL)
        stats = _ZERO_POOL.acquire(lib.bn_stats_floats(C), x.device) if want_stats else None
        PROFILE.bracket("dwconv_fwd", lambda: lib.dwconv2d_fwd(x, w, y, N, H, W, C, OH, OW, K, stride, pt, pl, stats=stats))
        ctx.geom = (N, H, W, C, OH, OW, K, stride, pt, pl)
        ctx.save_for_backward(x, w)
        ctx.w_param = w
        ctx.set_materialize_grads(False)
        if want_stats:
            ctx.mark_non_differentiable(stats)
            return y, stats
        return y

    @staticmethod
    def backward(ctx, gy, _gstats=None):
        if gy is None:
            return (None,) * 5
        x, w = ctx.saved_tensors
        lib = get_lib()
        gy = gy.contiguous(memory_format=_CL)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x, memory_format=_CL)
            PROFILE.bracket("dwconv_dgrad", lambda: lib.dwconv2d_bwd_data(gy, w, dx, *ctx.geom))
        if ctx.needs_input_grad[1]:
            p = ctx.w_param
            tgt = p.grad if (getattr(p, "_hifihr_direct_grad", False) and p.grad is not None and p.grad.is_contiguous()) else None
            if tgt is None:
                dw = torch.zeros_like(w)
                tgt = dw
            PROFILE.bracket("dwconv_wgrad", lambda: lib.dwconv2d_bwd_weight(x, gy, tgt, *ctx.geom))
            if dw is None:
                _grad_ready(p)
        return dx, dw, None, None, None


class _BNSwishDwConv(torch.autograd.Function):
    """depthwise_conv(swish(bn(e))) of an MBConv block (reference network/efficientnet_pt/model.py:73-80) WITHOUT the activated tensor:
    e = the expand convolution's raw output and `stats` its slot buffer.  Forward: hifihr_bn_finalize_fwd (one tiny launch: mean / invstd,
    running statistics) + hifihr_dwconv2d_fwd_bnswish (batch-norm + swish applied as the rows are loaded).  Backward: the weight gradient
    from e the same way; d a = dwconv_bwd_data(gy); then the ordinary fused batch-norm backward with act = swish on (d a, e), which
    recomputes the activation's derivative from e.  Against `bn_act` + `dwconv2d`: one pass over the block's largest tensor less in
    forward (read e + write a: 2.4 GB per EfficientNet-b3 step at batch 48) and that tensor is never allocated."""

    @staticmethod
    def forward(ctx, e, stats, gamma, beta, eps, momentum, running_mean, running_var, w, stride, pad4, want_stats):
        require_cuda(e, stats, gamma, beta, w)
        lib = get_lib()
        e = e.contiguous(memory_format=_CL)
        w = w.contiguous()
        N, C, H, W = e.shape
        K = w.shape[-1]
        pl, pr, pt, pb = pad4
        OH, OW = (H + pt + pb - K) // stride + 1, (W + pl + pr - K) // stride + 1
        save_mean, save_invstd = torch.empty(C, device=e.device), torch.empty(C, device=e.device)
        PROFILE.bracket("bn_fwd", lambda: lib.bn_finalize_fwd(stats, N * H * W, C, eps, momentum, save_mean, save_invstd, running_mean, running_var))
        _ZERO_POOL.release(stats)                 # consumed and zeroed
        y = torch.empty((N, C, OH, OW), device=e.device, dtype=torch.float32, memory_format=_CL)
        ystats = _ZERO_POOL.acquire(lib.bn_stats_floats(C), e.device) if want_stats else None
        geom = (N, H, W, C, OH, OW, K, stride, pt, pl)
        PROFILE.bracket("dwconv_fwd", lambda: lib.dwconv2d_fwd_bnswish(e, save_mean, save_invstd, gamma, beta, w, y, *geom, stats=ystats))
        ctx.geom, ctx.M, ctx.C = geom, N * H * W, C
        ctx.save_for_backward(e, w, gamma, beta, save_mean, save_invstd)
        ctx.w_param, ctx.gamma_param, ctx.beta_param = w, gamma, beta
        ctx.set_materialize_grads(False)
        if want_stats:
            ctx.mark_non_differentiable(ystats)
            return y, ystats
        return y

    @staticmethod
    def backward(ctx, gy, _gstats=None):
        if gy is None:
            return (None,) * 12
        e, w, gamma, beta, save_mean, save_invstd = ctx.saved_tensors
        lib = get_lib()
        gy = gy.contiguous(memory_format=_CL)
        dw = None
        if ctx.needs_input_grad[8]:
            p = ctx.w_param
            tgt = p.grad if (getattr(p, "_hifihr_direct_grad", False) and p.grad is not None and p.grad.is_contiguous()) else None
            if tgt is None:
                dw = torch.zeros_like(w)
                tgt = dw
            PROFILE.bracket("dwconv_wgrad", lambda: lib.dwconv2d_bwd_weight_bnswish(e, save_mean, save_invstd, gamma, beta, gy, tgt, *ctx.geom))
            if dw is None:
                _grad_ready(p)
        da = torch.empty_like(e, memory_format=_CL)
        PROFILE.bracket("dwconv_dgrad", lambda: lib.dwconv2d_bwd_data(gy, w, da, *ctx.geom))
        de = torch.empty_like(e, memory_format=_CL)
        red = _ZERO_POOL.acquire(lib.bn_stats_floats(ctx.C), e.device)

        def acc_target(p):
            if getattr(p, "_hifihr_direct_grad", False) and p.grad is not None:
                return p.grad, None
            t = torch.zeros(ctx.C, device=e.device)
            return t, t
        dg_t, dg_ret = acc_target(ctx.gamma_param)
        db_t, db_ret = acc_target(ctx.beta_param)
        PROFILE.bracket("bn_bwd", lambda: lib.bn_act_bwd(da, None, e, save_mean, save_invstd, gamma, beta, 2, ctx.M, ctx.C, red, de, None, dg_t, db_t))
        _ZERO_POOL.release(red)
        if dg_ret is None:
            _grad_ready(ctx.gamma_param)
        if db_ret is None:
            _grad_ready(ctx.beta_param)
        return de, None, dg_ret, db_ret, None, None, None, None, dw, None, None, None


def bn_swish_dwconv(e, stats, bn: torch.nn.BatchNorm2d, w, stride, pad4, want_stats=False):
    """dwconv2d(bn_act(e, stats, bn, None, "swish"), w, stride, pad4, want_stats) in training mode, without the activated tensor."""
    assert bn.training and stats is not None
    return _BNSwishDwConv.apply(e, stats, bn.weight, bn.bias, float(bn.eps), float(bn.momentum), bn.running_mean, bn.running_var, w, stride,
                                tuple(pad4), want_stats)


def dwconv2d(x, w, stride, pad4, want_stats=False):
    """F.conv2d(F.pad(x, pad4), w, groups=C, stride=stride) for channels_last fp32 tensors; pad4 = (left, right, top, bottom)
    (reference network/efficientnet_pt/utils.py:122-145 with groups = channels).  want_stats=True also returns the batch-norm
    statistics of the output (for `bn_act`), accumulated in the kernel's epilogue."""
    return _DwConv.apply(x, w, stride, tuple(pad4), want_stats)
