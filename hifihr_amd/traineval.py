"""The step of the reference's train loop (reference train_hrnet.py:50-113) and the batch plumbing around it
(reference utils/traineval_util.py:21-111 data_dic FreiHand branch, :338-354 trans_proj_j2d;
utils/fh_utils.py:30-39 proj_func)."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def proj_func(xyz, K):
    """fh_utils.py:30-39: uv = (K xyz)_xy / (K xyz)_z."""
    uv = (xyz.unsqueeze(2) * K.unsqueeze(1)).sum(3)          # [B,N,3]: row n = K . xyz_n, as broadcast multiply-adds (no BLAS call)
    return uv[:, :, :2] / uv[:, :, 2:3]


def trans_proj_j2d(outputs, Ks, root_xyz=None, which_joints="joints"):
    """traineval_util.py:338-354, the `scales is None` branch the step uses."""
    j3d = outputs[which_joints]
    if root_xyz is not None:
        j3d = j3d + root_xyz
    return proj_func(j3d, Ks)


_HO3D_TO_FREI = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]   # FreiHAND joint i <- HO-3D joint [i]


def HO3D2Frei(ho3d_joints):
    """utils/fh_utils.py:604-616: HO-3D joint order -> FreiHAND joint order (one gather instead of 21 slice copies)."""
    return ho3d_joints[:, _HO3D_TO_FREI]


def Frei2HO3D(frei_joints):
    """utils/fh_utils.py:618-629 (inverse permutation; the HO-3D evaluation dump uses it)."""
    inv = [0] * 21
    for frei, ho in enumerate(_HO3D_TO_FREI):
        inv[ho] = frei
    return frei_joints[:, inv]


_RHD_TO_FREI = [0, 4, 3, 2, 1, 8, 7, 6, 5, 12, 11, 10, 9, 16, 15, 14, 13, 20, 19, 18, 17]       # FreiHAND joint i <- RHD joint [i]


def RHD2Frei(rhd_joints):
    """utils/fh_utils.py:590-602: RHD joint order (wrist, then tip -> base per finger) -> FreiHAND order, as one gather."""
    return rhd_joints[:, _RHD_TO_FREI]


def data_dic(sample, dat_name, set_name, args, device="cuda", image_size=224):
    """utils/traineval_util.py:21-111 (FreiHand branch; training queries [trans_images, trans_Ks, trans_joints, scales,
    trans_verts, trans_masks], evaluation queries without the prefix) and :156-201 (HO3D branch: crop resized to 224 with
    nearest-neighbour `interpolate`, the OpenGL -> image convention flip of K's columns and of the joints' y / z by
    (1, -1, -1), HO-3D -> FreiHAND joint order).  Host tensors go to `device`; `Ps = Ks . [I|0]` is a zero column."""
    ex = {}
    to = lambda t: t.to(device, non_blocking=True)
    if dat_name == "FreiHand":
        training = "training" in set_name
        g = lambda k: sample["trans_" + k] if ("trans_" + k) in sample and (training or k in ("images", "Ks")) else sample.get(k)
        ex["imgs"] = to(g("images"))
        Ks = to(g("Ks"))
        if "scales" in sample:
            ex["scales"] = to(sample["scales"].float())
        ex["idxs"] = to(sample["idxs"])
        joints, verts, masks = g("joints"), g("verts"), g("masks")
    elif dat_name == "HO3D":
        # the reference resizes the crop to 224 (its render resolution is hard-wired to 224); `image_size` is the build-side
        # render resolution of BASELINE configs[4] (512): the photometric terms then run at that size, the encoder reads a 224 resize
        ex["imgs"] = F.interpolate(to(sample["img_crop"]), (image_size, image_size))
        flip = torch.tensor([1.0, -1.0, -1.0], device=device)
        Ks = to(sample["K_crop"]) * flip.view(1, 1, 3)
        if "root_xyz" in sample:
            ex["root_xyz"] = to(sample["root_xyz"])
        if "uv21_crop" in sample:
            ex["j2d_gt"] = HO3D2Frei(to(sample["uv21_crop"].float()))
        joints = HO3D2Frei(to(sample["xyz21"])) * flip.view(1, 1, 3) if "xyz21" in sample else None
        verts, masks = None, sample.get("hand_mask_crop")
        if masks is not None and masks.shape[-1] != image_size:             # the loader's 224 crop at a build-side render resolution
            masks = F.interpolate(to(masks), (image_size, image_size))
    elif dat_name == "RHD":
        # utils/traineval_util.py:204-256: cropped image and intrinsics as loaded, 2-D / 3-D joints re-ordered RHD -> FreiHAND,
        # keypoint_scale doubles as `scales`, visibility flags re-ordered; no vertices, no masks (the mask lines are commented out there)
        ex["imgs"] = to(sample["img_crop"])
        Ks = to(sample["K_crop"])
        if "uv21_crop" in sample:
            ex["j2d_gt"] = RHD2Frei(to(sample["uv21_crop"]))
        joints = RHD2Frei(to(sample["xyz21"])) if "xyz21" in sample else None
        if "keypoint_scale" in sample:
            ex["keypoint_scale"] = to(sample["keypoint_scale"])
            ex["scales"] = ex["keypoint_scale"]
        if "uv_vis" in sample:
            ex["uv_vis"] = RHD2Frei(to(sample["uv_vis"]))           # on the device like every other entry (a captured step copies them all)
        if "open_2dj" in sample:
            ex["open_2dj"], ex["open_2dj_con"] = to(sample["open_2dj_crop"]), to(sample["open_2dj_con"])
        verts, masks = None, None
    else:
        raise NotImplementedError(f"dat_name='{dat_name}': FreiHand, HO3D and RHD are built (Obman / Dart need their datasets)")
    ex["Ks"] = Ks
    ex["Ps"] = torch.cat([Ks, torch.zeros_like(Ks[:, :, :1])], dim=2)       # Ks @ [I|0]
    if joints is not None:
        ex["joints"] = to(joints)
        if dat_name == "FreiHand":                 # (the RHD branch's projected 2-D joints are commented out in the reference, :239)
            ex["j2d_gt"] = proj_func(ex["joints"], Ks)
    if verts is not None:
        ex["verts"] = to(verts)
    if masks is not None:
        ex["masks"] = to(masks)
        ex["segms_gt"] = ex["masks"][:, 0].long()
    return ex


_BACKWARD_SEED = {}


def _backward_seed(loss):
    """d loss / d loss = 1 as a constant kept per device (autograd's own ones_like is a fill launch per step).  Never CREATED inside a
    stream capture: there the fill would only be recorded and the tensor would live in that graph's private pool -- later graphs and
    eager steps would read memory that holds 1.0 only after the first graph has replayed.  The steppers create it before they capture
    (`_prime_for_capture`); a capture that still finds the cache empty gets autograd's default seed (None)."""
    seed = _BACKWARD_SEED.get(loss.device)
    if seed is None or seed.dtype != loss.dtype or seed.shape != loss.shape:
        if torch.cuda.is_current_stream_capturing():
            return None
        seed = _BACKWARD_SEED[loss.device] = torch.ones_like(loss)
    return seed


def _prime_for_capture(device):
    """What a capture must find ready (neither may be allocated while capturing): the backward seed and the library's zero page."""
    from ._lib import get_lib
    dev = torch.device(device)
    if dev not in _BACKWARD_SEED:
        _BACKWARD_SEED[dev] = torch.ones((), device=dev, dtype=torch.float32)
    get_lib().zero_page_ready(dev)


def forward_backward(model, loss_func, optimizer, examples, args, dat_name="FreiHand"):
    """Forward, losses, zero_grad and backward of one iteration (train_hrnet.py:50-104).  Returns (loss, loss_dic)."""
    from .ops import prepared_weights
    # one launch re-lays every convolution weight out for this step (ops._WeightPrep); weight gradients go to a side stream
    # (ops._AsyncWgrad) unless data-parallel hooks may launch a bucket's all-reduce in the middle of backward
    dp = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
    with prepared_weights(async_wgrad=not dp):
        return _forward_backward(model, loss_func, optimizer, examples, args, dat_name)


def _forward_backward(model, loss_func, optimizer, examples, args, dat_name):
    # batches assembled by data.FreiHandDeviceCache.batch_examples(root_id=args.ROOT) carry the terms derived here on every iteration
    # (root_xyz, the root-relative ground truth, the NDC camera): four elementwise launches fewer per step
    pre = dat_name != "HO3D" and "joints_rel" in examples and "root_xyz" in examples
    root_xyz = examples["root_xyz"] if pre else examples["joints"][:, args.ROOT, :].unsqueeze(1)
    if pre and "verts" in examples and "verts_rel" not in examples:
        # a hand-built dict with some of the step terms only: the vertex ground truth must be root-relative like the joints'
        examples = dict(examples, verts_rel=examples["verts"] - root_xyz)
    # the batch kernel's NDC camera is scaled with the CACHE's image size (= the size of `imgs` it produced); Model.camera_from_K uses the
    # model's.  When the two differ the precomputed camera is not the model's own: fall back to camera_from_K(Ks)
    same_size = getattr(model, "image_size", None) in (None, examples["imgs"].shape[-1])
    kw = {"cam_ndc": examples["cam_ndc"]} if ("cam_ndc" in examples and getattr(model, "accepts_cam_ndc", False) and same_size) else {}
    outputs = model(dat_name, True, examples["imgs"], Ks=examples["Ps"], root_xyz=root_xyz, **kw)
    ex = dict(examples)
    if pre:
        ex["joints"] = examples["joints_rel"]
        if "verts_rel" in examples:
            ex["verts"] = examples["verts_rel"]
    elif dat_name != "HO3D":                 # train_hrnet.py:64-68: HO-3D keeps its absolute ground truth
        ex["joints"] = examples["joints"] - root_xyz
        if "verts" in examples:
            ex["verts"] = examples["verts"] - root_xyz
    if any(k in args.losses for k in ("joint_2d", "bone_direc")):
        outputs["j2d"] = trans_proj_j2d(outputs, examples["Ks"], root_xyz=root_xyz)      # only these terms read it
    loss_dic = loss_func(ex, outputs, args.losses, dat_name, args)
    missing = [k for k in args.losses if k not in loss_dic]
    if missing:                              # e.g. 'mtex' on a hand layer without texture_params, 'vert_3d' on a dataset without vertices
        raise KeyError(f"loss terms {missing} were requested but not produced for {dat_name}: their inputs are absent from the model outputs / examples")
    if hasattr(loss_func, "total"):
        loss = loss_func.total(loss_dic, args.losses)          # one launch when the terms sit in the fused kernels' vectors
    else:
        terms = [loss_dic[k] for k in args.losses]
        loss = terms[0] if len(terms) == 1 else torch.stack(terms).sum()      # 2 launches instead of a chain of adds
    loss_dic["loss"] = loss
    optimizer.zero_grad(set_to_none=True)
    loss.backward(_backward_seed(loss))
    return loss, loss_dic


def train_step(model, loss_func, optimizer, examples, args, dat_name="FreiHand", backward_hook=None):
    """One iteration of train_an_epoch (train_hrnet.py:50-113), mode_train=True.  Returns (loss, loss_dic)."""
    loss, loss_dic = forward_backward(model, loss_func, optimizer, examples, args, dat_name)
    if backward_hook is not None:
        backward_hook()                      # data-parallel gradient all-reduce (hifihr_amd/dist.py)
    optimizer.step()
    return loss, loss_dic


def _check_step_terms(static, examples):
    """A captured step whose inputs include the batch-derived terms (root_xyz, joints_rel, verts_rel, cam_ndc: data.FreiHandDeviceCache.
    batch_examples(root_id=...)) reads THEM, not joints / Ks: a batch that arrives without them would leave the previous batch's in place."""
    missing = [k for k in ("root_xyz", "joints_rel", "verts_rel", "cam_ndc") if k in static and k not in examples]
    if missing:
        raise KeyError(f"load_batch: the captured step was built on a batch that carries {missing}; pass batches from the same source "
                       f"(batch_examples(..., root_id=args.ROOT)) or capture the step on a batch without them")


def _injected_capture_failure():
    """Test hook (tests/test_gpu_dp.py): HIFIHR_TEST_FAIL_CAPTURE_RANK=<r> makes the graphed steps' constructors fail on rank r only, to
    exercise the collective-safe fallback -- a one-sided capture failure must send every rank to the eager step."""
    import os
    r = os.environ.get("HIFIHR_TEST_FAIL_CAPTURE_RANK")
    if r is not None and torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_rank() == int(r):
        raise RuntimeError("injected capture failure (HIFIHR_TEST_FAIL_CAPTURE_RANK)")


class GraphedTrainStep:
    """The training iteration captured once into a hipGraph and replayed: ~350 kernel launches per step collapse into one
    graph launch, which removes the host-side launch gaps (the eager step is launch-bound in places: DESIGN.md section 6).
    The batch lives in static device tensors that `load_batch` overwrites in place.

    Single process (`reducer=None`): forward, losses, backward AND the fused Adam are in the graph; the per-step Adam
    scalars are refreshed outside it (FusedAdam.prepare_step).
    Data parallel (`reducer` = hifihr_amd.dist.GradReducer): the graph ends after backward; the flat gradient buffer is
    then all-reduced (bucketed, asynchronous, RCCL) and the fused Adam step is launched eagerly -- no collective is ever
    captured, each rank replays its own graph.  The exchange is not overlapped with backward in this mode (about 0.4 ms
    for 47 MB over xGMI against the ~3.5 ms the launch gaps of the eager step cost).

    The model must never have run a step on the legacy default stream (autograd pins gradient accumulation to the
    stream of first use): callers do `torch.cuda.set_stream(torch.cuda.Stream())` before the first step."""

    def __init__(self, model, loss_func, optimizer, examples, args, dat_name="FreiHand", warmup=3, reducer=None):
        self.model, self.loss_func, self.opt, self.args, self.dat_name = model, loss_func, optimizer, args, dat_name
        self.reducer = reducer
        self.static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in examples.items()}
        split = reducer is not None
        # Warm-up and capture must leave the training state exactly as they found it (the warm-up runs real steps on one batch;
        # a re-capture happens whenever a lambda schedule fires): parameters, Adam moments, step counter and every module buffer
        # (batch-norm running statistics) are snapshotted here and restored below.
        flat = optimizer.flatp
        snap = (flat.flat.clone(), optimizer.exp_avg.clone(), optimizer.exp_avg_sq.clone(), optimizer.step_count,
                [b.clone() for b in model.buffers()])
        if split:
            reducer.pause_hooks(True)                           # the graph must not contain (or trigger) collectives
        else:
            optimizer.enable_graph_mode()
        try:
            _injected_capture_failure()
            _prime_for_capture(flat.flat.device)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                       # warm-up on a side stream (allocator / workspace growth)
                for _ in range(warmup):
                    if split:
                        # NO collective in the warm-up (nor in the capture): a rank whose warm-up or capture raises can then meet the
                        # others in the caller's agreement all-reduce, the first collective after this constructor -- with the bucket
                        # all-reduces here, a one-sided failure left the other ranks inside collectives the failing rank never joined.
                        # The state is restored below, so ranks stepping on local gradients meanwhile is harmless.
                        forward_backward(model, loss_func, optimizer, self.static, args, dat_name)
                        optimizer.step()
                    else:
                        optimizer.prepare_step()
                        train_step(model, loss_func, optimizer, self.static, args, dat_name)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            if split:
                # thread-local capture: the process-group watchdog thread queries events while we capture; in the default (global)
                # mode any such call from another thread invalidates the capture
                with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                    self.loss, self.loss_dic = forward_backward(model, loss_func, optimizer, self.static, args, dat_name)
            else:
                optimizer.prepare_step()
                with torch.cuda.graph(self.graph):
                    self.loss, self.loss_dic = train_step(model, loss_func, optimizer, self.static, args, dat_name)
        except BaseException:
            # leave the optimizer / reducer usable by the eager step (a half-enabled graph mode froze lr and bias correction)
            if split:
                reducer.pause_hooks(False)
            else:
                optimizer.disable_graph_mode()
            self._restore(snap)
            raise
        self._restore(snap)

    def _restore(self, snap):
        flat_w, m, v, step, bufs = snap
        with torch.no_grad():
            self.opt.flatp.flat.copy_(flat_w)
            self.opt.exp_avg.copy_(m)
            self.opt.exp_avg_sq.copy_(v)
            for b, s in zip(self.model.buffers(), bufs):
                b.copy_(s)
        self.opt.step_count = step
        torch.cuda.synchronize()

    def release(self):
        """Hand the optimizer / reducer back to the eager step (train_hrnet.py drops a stepper when a lambda schedule fires)."""
        if self.reducer is not None:
            self.reducer.pause_hooks(False)
        else:
            self.opt.disable_graph_mode()

    def load_batch(self, examples):
        _check_step_terms(self.static, examples)
        for k, v in examples.items():
            if torch.is_tensor(v):
                self.static[k].copy_(v, non_blocking=True)

    def __call__(self):
        if self.reducer is not None:
            self.graph.replay()
            self.reducer.all_reduce_flat()
            self.opt.step()
        else:
            self.opt.prepare_step()
            self.graph.replay()
            self.opt.note_step_done()
        return self.loss, self.loss_dic


class SegmentedGraphedTrainStep:
    """Data-parallel step whose gradient exchange OVERLAPS backward although every launch is a graph replay (round 3; review item 8).

    The autograd graph of the ResNet trunk is cut at the layer boundaries (network.Resnet_4C.segment_cut: the boundary tensor is
    detached and re-enters as a leaf), which makes backward a sequence of independent calls -- loss.backward() stops at the layer-4
    output, layer4_out.backward(grad) at the layer-3 output, ... -- and each call is captured into a hipGraph of its own (one shared
    memory pool).  A step is then: replay forward; replay backward segment k; enqueue the all-reduce of the parameters segment k
    completed (RCCL runs it on its own stream, ordered behind the replay just enqueued); replay segment k - 1 meanwhile; ...; wait;
    fused Adam.  Graph BRANCHES were measured to cost more than they hide on this runtime: the overlap comes
    from separate launches, no collective is captured.

    Segments (parameters in registration order): [mmpool .. layer2] (exchanged last), [layer3], [layer4], [heads: hand encoder, light
    estimator] (first).  Same arithmetic as GraphedTrainStep(reducer=...): the kernels and their order inside a segment are unchanged
    (only layer3 -> layer4 hands its output over materialised instead of un-normalised)."""

    def __init__(self, model, loss_func, optimizer, examples, args, reducer, dat_name="FreiHand", warmup=3):
        enc = getattr(getattr(model, "base_encoder", None), "encoder1", None)
        if enc is None or not hasattr(enc, "model") or not hasattr(enc.model, "layer3"):
            raise NotImplementedError("segmented backward is built for the ResNet trunks")
        self.model, self.loss_func, self.opt, self.args, self.dat_name, self.reducer = model, loss_func, optimizer, args, dat_name, reducer
        self.static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in examples.items()}
        flat = optimizer.flatp
        idx = {id(p): i for i, p in enumerate(flat.params)}

        def span(mods):
            ids = [idx[id(p)] for m in mods for p in m.parameters() if id(p) in idx]
            return (min(ids), max(ids) + 1) if ids else (0, 0)
        trunk = enc.model
        lo3, hi3 = span([trunk.layer3])
        lo4, hi4 = span([trunk.layer4])
        n = len(flat.params)
        # exchanged after segment:     heads            layer4       layer3       stem .. layer2 (+ whatever precedes it in the buffer)
        self.ranges = [(hi4, n), (lo4, hi4), (lo3, hi3), (0, lo3)]
        assert 0 < lo3 <= hi3 == lo4 <= hi4 <= n, (lo3, hi3, lo4, hi4, n)
        snap = (flat.flat.clone(), optimizer.exp_avg.clone(), optimizer.exp_avg_sq.clone(), optimizer.step_count,
                [b.clone() for b in model.buffers()])
        reducer.pause_hooks(True)
        self._enc = enc
        self._scope = None
        try:
            _injected_capture_failure()                        # (test hook: the one-sided failure of tests/test_gpu_dp.py)
            _prime_for_capture(flat.flat.device)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):                        # (collective-free, like GraphedTrainStep's: see there)
                    self._eager_segmented()
                    optimizer.step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            pool = torch.cuda.graph_pool_handle()
            self.graphs = []
            stages = self._stages()
            for stage in stages:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                    stage()
                self.graphs.append(g)
        except BaseException:
            reducer.pause_hooks(False)
            enc.segment_cut = None
            if self._scope is not None:                        # a stage raised between `fwd` (which enters the weight re-layout scope) and
                self._scope.__exit__(None, None, None)         # `bwd_last` (which leaves it): without this _WEIGHT_PREP stays active and a
                self._scope = None                             # later forward outside any scope reads last step's re-laid-out weights
            self._restore(snap)
            raise
        enc.segment_cut = None
        self._restore(snap)

    # -- the step as five callables: forward (+ zero_grad), then the four backward segments; state shared through self._b
    def _stages(self):
        from .ops import _WEIGHT_PREP, prepared_weights

        def fwd():
            self._b = {}

            def cut(name, t):
                leaf = t.detach().requires_grad_(True)
                self._b[name] = (t, leaf)
                return leaf
            self._enc.segment_cut = cut
            # the weight re-layout scope spans all five stages: entered here, left after the last backward segment
            self._scope = prepared_weights(async_wgrad=False)
            self._scope.__enter__()
            root_xyz = self.static["joints"][:, self.args.ROOT, :].unsqueeze(1)
            self.loss, self.loss_dic = _forward_only(self.model, self.loss_func, self.opt, self.static, self.args, self.dat_name, root_xyz)

        def bwd_heads():
            self.loss.backward()

        def bwd(name):
            def run():
                t, leaf = self._b[name]
                t.backward(leaf.grad)
            return run

        def bwd_last():
            bwd("layer2")()
            self._scope.__exit__(None, None, None)
            self._scope = None
        return [fwd, bwd_heads, bwd("layer4"), bwd("layer3"), bwd_last]

    def _eager_segmented(self):
        for stage in self._stages():
            stage()

    def _restore(self, snap):
        GraphedTrainStep._restore(self, snap)

    def release(self):
        self.reducer.pause_hooks(False)

    def load_batch(self, examples):
        _check_step_terms(self.static, examples)
        for k, v in examples.items():
            if torch.is_tensor(v):
                self.static[k].copy_(v, non_blocking=True)

    def __call__(self):
        self.graphs[0].replay()
        handles = []
        for g, (lo, hi) in zip(self.graphs[1:], self.ranges):
            g.replay()
            h = self.reducer.all_reduce_params_async(lo, hi)
            if h is not None:
                handles.append(h)
        for h in handles:
            h.wait()
        self.opt.step()
        return self.loss, self.loss_dic


def _forward_only(model, loss_func, optimizer, examples, args, dat_name, root_xyz):
    """_forward_backward without the backward call (the segmented step runs backward in pieces)."""
    outputs = model(dat_name, True, examples["imgs"], Ks=examples["Ps"], root_xyz=root_xyz)
    ex = dict(examples)
    if dat_name != "HO3D":
        ex["joints"] = examples["joints"] - root_xyz
        if "verts" in examples:
            ex["verts"] = examples["verts"] - root_xyz
    if any(k in args.losses for k in ("joint_2d", "bone_direc")):
        outputs["j2d"] = trans_proj_j2d(outputs, examples["Ks"], root_xyz=root_xyz)
    loss_dic = loss_func(ex, outputs, args.losses, dat_name, args)
    missing = [k for k in args.losses if k not in loss_dic]
    if missing:
        raise KeyError(f"loss terms {missing} were requested but not produced for {dat_name}")
    terms = [loss_dic[k] for k in args.losses]
    loss = terms[0] if len(terms) == 1 else torch.stack(terms).sum()
    loss_dic["loss"] = loss
    optimizer.zero_grad(set_to_none=True)
    return loss, loss_dic
