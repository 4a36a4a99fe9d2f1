"""The reference's `.t7` checkpoint (reference utils/train_utils.py:14-113 load_model, :116-202 save_model).

A `.t7` file is a `torch.save`d dict:
    {'args': Namespace, 'optimizer': torch.optim.Adam.state_dict(), 'scheduler': MultiStepLR.state_dict(), 'epoch': int,
     'base_encoder': ..., 'hand_encoder': ..., 'light_estimator': ...}            (one state dict per sub-module)

This build keeps the reference's module structure, so the sub-module state dicts carry the same names and logical
[K,C,R,S] / [out,in] shapes (tests/golden/state_dict_names.json pins them); conv weights live in channels_last memory
here and are written out contiguous.  Two differences are bridged:
  * the reference's encoders carry the classifier head of their torchvision / EfficientNet parent (`encoder1.model.fc.*`,
    `encoder._fc.*`), which no forward uses.  Loading drops them (and remembers them); saving writes them back (or a
    zero head) so the reference's strict `load_state_dict` accepts the file.
  * the optimizer state.  The reference holds a torch.optim.Adam over `model.parameters()` (train_hrnet.py:546-550), i.e.
    per-parameter `exp_avg / exp_avg_sq / step` keyed by the parameter's position -- classifier heads included.  Here
    the moments are two flat buffers (hifihr_amd/optim.FusedAdam); they are cut into / assembled from that format.
"""
from __future__ import annotations

import os

import torch

_SUBMODULES = ("base_encoder", "hand_encoder", "light_estimator")
_HEADS = {"encoder1.model.fc": (1000, 512), "encoder._fc": (1000, 1536)}          # unused classifier heads of the reference encoders
#                                                                                   (512 = ResNet-18; 2048 for ResNet-50 / -101: _head_shape)


def _head_shape(head, base_encoder_state):
    o, i = _HEADS[head]
    if head == "encoder1.model.fc" and any(".bn3." in k for k in base_encoder_state):     # bottleneck trunk
        i = 2048
    return o, i


def _head_prefix(base_encoder_state):
    return "encoder._fc" if any(k.startswith("encoder._") for k in base_encoder_state) else "encoder1.model.fc"


def reference_param_names(model):
    """Names of `reference_model.parameters()` in order: this model's trainable parameters with the unused classifier head
    (weight, bias) right after the image encoder's own parameters."""
    names = [n for n, p in model.named_parameters()]
    enc = [n for n in names if n.startswith("base_encoder.")]
    head = "base_encoder." + _head_prefix([n[len("base_encoder."):] for n in enc])
    i = names.index(enc[-1]) + 1
    return names[:i] + [head + ".weight", head + ".bias"] + names[i:]


def adam_state_to_torch(model, opt):
    """FusedAdam -> torch.optim.Adam.state_dict() as the reference's optimizer would hold it."""
    flat = opt.flatp
    by_param = {id(p): o for p, o in zip(flat.params, flat.offsets)}
    named = dict(model.named_parameters())
    state, names = {}, reference_param_names(model)
    for idx, n in enumerate(names):
        p = named.get(n)
        if p is None or id(p) not in by_param or opt.step_count == 0:
            continue                                   # classifier head / frozen: Adam never created state for it
        o = by_param[id(p)]
        cut = lambda buf: flat._view(buf, p, o).detach().clone().contiguous()
        state[idx] = {"step": torch.tensor(float(opt.step_count)), "exp_avg": cut(opt.exp_avg), "exp_avg_sq": cut(opt.exp_avg_sq)}
    g = opt.param_groups[0]
    group = {"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": g["weight_decay"], "amsgrad": False,
             "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
             "params": list(range(len(names)))}
    if "initial_lr" in g:
        group["initial_lr"] = g["initial_lr"]
    return {"state": state, "param_groups": [group]}


def adam_state_from_torch(model, opt, sd):
    """torch.optim.Adam.state_dict() (reference layout) -> FusedAdam.  Parameters without state keep zero moments."""
    flat = opt.flatp
    by_param = {id(p): o for p, o in zip(flat.params, flat.offsets)}
    named = dict(model.named_parameters())
    names = reference_param_names(model)
    order = sd["param_groups"][0]["params"]
    if len(order) != len(names):
        raise ValueError(f"optimizer state covers {len(order)} parameters, this model (with the classifier head) has {len(names)}")
    steps = set()
    opt.exp_avg.zero_(); opt.exp_avg_sq.zero_()
    for pos, key in enumerate(order):
        st = sd["state"].get(key)
        p = named.get(names[pos])
        if st is None or p is None or id(p) not in by_param:
            continue
        o = by_param[id(p)]
        if tuple(st["exp_avg"].shape) != tuple(p.shape):
            raise ValueError(f"optimizer state of {names[pos]}: shape {tuple(st['exp_avg'].shape)} vs parameter {tuple(p.shape)}")
        flat._view(opt.exp_avg, p, o).copy_(st["exp_avg"])
        flat._view(opt.exp_avg_sq, p, o).copy_(st["exp_avg_sq"])
        steps.add(int(float(st["step"])))
    if len(steps) > 1:
        raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): the fused Adam keeps one")
    opt.step_count = steps.pop() if steps else 0
    g, src = opt.param_groups[0], sd["param_groups"][0]
    for k in ("lr", "betas", "eps", "weight_decay", "initial_lr"):
        if k in src:
            g[k] = tuple(src[k]) if k == "betas" else src[k]


def model_state(model):
    """{'base_encoder': ..., 'hand_encoder': ..., 'light_estimator': ...} with the reference's strict key sets."""
    out = {}
    extra = getattr(model, "_reference_extra_state", {})
    for sub in _SUBMODULES:
        if not hasattr(model, sub):
            continue
        sd = {k: v.detach().clone().contiguous() for k, v in getattr(model, sub).state_dict().items()}
        if sub == "base_encoder":
            head = _head_prefix(sd)
            o, i = _head_shape(head, sd)
            sd[head + ".weight"] = extra.get(head + ".weight", torch.zeros(o, i))
            sd[head + ".bias"] = extra.get(head + ".bias", torch.zeros(o))
        out[sub] = sd
    return out


def save_model(model, optimizer, scheduler, epoch, current_epoch, args, console=None):
    """utils/train_utils.py:116-202 for task 'train': writes <state_output>/texturehand_<postfix>.t7 (postfix = epoch number
    with save_mode 'separately', 'latest' with 'only_latest'; an extra numbered copy every 20 epochs)."""
    state = {"args": args, "optimizer": adam_state_to_torch(model, optimizer), "epoch": epoch + current_epoch,
             "scheduler": scheduler.state_dict() if scheduler is not None else {}}
    state.update(model_state(model))
    postfix = epoch + current_epoch if getattr(args, "save_mode", "only_latest") == "separately" else "latest"
    os.makedirs(args.state_output, exist_ok=True)
    files = [os.path.join(args.state_output, f"texturehand_{postfix}.t7")]
    if (epoch + current_epoch) % 20 == 0:
        files.append(os.path.join(args.state_output, f"texturehand_{epoch + current_epoch}.t7"))
    for f in dict.fromkeys(files):
        torch.save(state, f)
        if console is not None:
            console.log(f"Save model at {f}")
    return files


def load_model(model, optimizer, scheduler, args):
    """utils/train_utils.py:14-113: restores the sub-modules present in the file, then optimizer and scheduler, and returns
    (model, current_epoch, optimizer, scheduler) like the reference.  `args.pretrain_model` is the .t7 path (None: nothing
    to do).  An optimizer state that does not fit is reported and skipped, as the reference's try/except does (:88-91)."""
    path = getattr(args, "pretrain_model", None)
    if path is None:
        return model, 0, optimizer, scheduler
    sd = torch.load(path, map_location="cpu", weights_only=False)       # the file pickles an argparse.Namespace
    extra = {}
    for sub in _SUBMODULES:
        if sub not in sd or not hasattr(model, sub):
            continue
        part = dict(sd[sub])
        if sub == "base_encoder":
            for head in _HEADS:
                for leaf in (".weight", ".bias"):
                    if head + leaf in part:
                        extra[head + leaf] = part.pop(head + leaf)
        getattr(model, sub).load_state_dict(part, strict=True)
    model._reference_extra_state = extra
    if optimizer is not None and sd.get("optimizer"):
        try:
            adam_state_from_torch(model, optimizer, sd["optimizer"])
        except ValueError as e:
            print("optimizer not loaded:", e)
    if scheduler is not None and sd.get("scheduler"):
        scheduler.load_state_dict(sd["scheduler"])
    return model, int(sd.get("epoch", 0)), optimizer, scheduler


def rec_freeze(module):
    """utils/visualize_util.py:932-939: batch-norm momentum 0 everywhere below `module` (running statistics stop moving; batch
    statistics are still used in train mode) and requires_grad off for every parameter that lives in a CHILD of `module`
    (parameters held directly by `module` itself keep their flag -- the reference's recursion has that quirk)."""
    for m in module.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.momentum = 0
    for child in module.children():
        for p in child.parameters():
            p.requires_grad = False
        rec_freeze(child)


def freeze_model_modules(model, args):
    """utils/train_utils.py:205-240 for the sub-modules this build has.  Call BEFORE optim.FlatParams is built: frozen parameters
    then stay out of the flat buffers, get no gradient kernels (the ops look at needs_input_grad) and no Adam update."""
    frozen = []
    if getattr(args, "only_train_regressor", False) and hasattr(model, "light_estimator"):
        rec_freeze(model.light_estimator); frozen.append("light_estimator")
    if getattr(args, "only_train_texture", False):
        if hasattr(model, "base_encoder"):
            rec_freeze(model.base_encoder); frozen.append("base_encoder")
        for name in ("base_layers", "pose_reg", "shape_reg"):
            rec_freeze(getattr(model.hand_encoder, name)); frozen.append("hand_encoder." + name)
    return frozen
