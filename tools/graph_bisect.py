"""Capture individual hifihr ops in a hipGraph (each in its own subprocess) to find what breaks capture."""
import subprocess, sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["fwd_loss", "fwd_loss_bwd", "fwd_loss_bwd_flat", "full_step", "adam", "mano", "render_fwd", "render_bwd", "ssim", "conv", "torch_bn", "encoder_fwd", "encoder_fwdbwd", "memset_only", "full_nograd"]
TEMPLATE = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import torch
from hifihr_amd._lib import get_lib
from hifihr_amd.mano_tables import synthetic_mano_tables
from hifihr_amd import ops
import kernel_cases as kc
case = %r
lib = get_lib(); t = synthetic_mano_tables(0); dev = "cuda"
B = 8
def run_capture(fn, nwarm=2):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(nwarm): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    g.replay(); g.replay(); torch.cuda.synchronize()
    return out
if case == "adam":
    p = torch.randn(100003, device=dev); gr = torch.randn_like(p); m = torch.zeros_like(p); v = torch.zeros_like(p); dyn = torch.tensor([1e-3, 1.0], device=dev)
    run_capture(lambda: lib.adam_step_dyn(p, gr, m, v, 1.0, 0.9, 0.999, 1e-8, 0.0, dyn)); print("ok", float(p.sum()))
elif case == "mano":
    h = ops.ManoLayerHandle(t); pose = (0.3 * torch.randn(B, 48, device=dev)).requires_grad_(True); beta = torch.zeros(B, 10, device=dev, requires_grad=True)
    def f():
        v, j = ops.mano_lbs(h, pose, beta); jr, vr, r = ops.mano_joints_root_relative(h, v, 9)
        pose.grad = None; (vr.sum() + jr.sum()).backward(); return pose.grad
    o = run_capture(f); print("ok", float(o.abs().sum()))
elif case in ("render_fwd", "render_bwd"):
    verts, vcol, cam, lc, ld = (x.cuda().contiguous() for x in kc.make_render_inputs(t, B, 7, 224))
    r = ops.RendererHandle(t.faces, 778)
    if case == "render_fwd":
        o = run_capture(lambda: ops.render(r, verts, vcol, cam, lc, ld)[0]); print("ok", float(o.mean()))
    else:
        verts.requires_grad_(True)
        def f():
            verts.grad = None; ops.render(r, verts, vcol, cam, lc, ld)[0][:, :3].sum().backward(); return verts.grad
        o = run_capture(f); print("ok", float(o.abs().sum()))
elif case == "ssim":
    a = torch.rand(B, 3, 224, 224, device=dev, requires_grad=True); b = torch.rand(B, 3, 224, 224, device=dev)
    def f():
        a.grad = None; ops.ssim(a, b).backward(); return a.grad
    o = run_capture(f); print("ok", float(o.abs().sum()))
elif case == "conv":
    x = torch.randn(B, 64, 56, 56, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (0.05 * torch.randn(64, 64, 3, 3, device=dev)).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    def f():
        x.grad = None; w.grad = None; ops.conv2d(x, w, 1, 1).sum().backward(); return w.grad
    o = run_capture(f); print("ok", float(o.abs().sum()))
elif case == "torch_bn":
    bn = torch.nn.BatchNorm2d(64).cuda(); x = torch.randn(B, 64, 56, 56, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    def f():
        x.grad = None; torch.relu(bn(x)).sum().backward(); return x.grad
    o = run_capture(f); print("ok", float(o.abs().sum()))
elif case in ("encoder_fwd", "encoder_fwdbwd"):
    from hifihr_amd.network import ResEncoder
    enc = ResEncoder(conv_impl="mfma").cuda().train(); img = torch.rand(B, 3, 224, 224, device=dev)
    def f():
        low, feat = enc(img)
        if case == "encoder_fwdbwd":
            for p in enc.parameters(): p.grad = None
            (low.sum() + feat.sum()).backward()
        return feat
    o = run_capture(f); print("ok", float(o.abs().sum()))
elif case == "memset_only":
    z = torch.empty(1000, device=dev)
    o = run_capture(lambda: z.zero_()); print("ok")
elif case in ("fwd_loss", "fwd_loss_bwd", "fwd_loss_bwd_flat", "full_step"):
    from hifihr_amd.models import Model
    from hifihr_amd import options, synth
    from hifihr_amd.losses import LossFunction
    from hifihr_amd.optim import FlatParams, FusedAdam
    from hifihr_amd.traineval import data_dic, train_step, trans_proj_j2d
    args = options.baseline_config2_args(train_batch=B)
    model = Model(True, torch.device(dev), False, "mano", False, "res18", mano_tables=t).cuda().train()
    ex = data_dic(synth.make_batch(model.hand_layer.handle, model.renderer_p3d, B, device=dev), "FreiHand", "training", args, device=dev)
    root = ex["joints"][:, 9, :].unsqueeze(1)
    lf = LossFunction()
    if case in ("fwd_loss_bwd_flat", "full_step"):
        flat = FlatParams(model); opt = FusedAdam(flat, lr=1e-6); opt.enable_graph_mode(); opt.prepare_step()
    def f():
        if case == "full_step":
            return train_step(model, lf, opt, ex, args)[0]
        out = model("FreiHand", True, ex["imgs"], Ks=ex["Ps"], root_xyz=root)
        e2 = dict(ex); e2["joints"] = ex["joints"] - root; e2["verts"] = ex["verts"] - root
        out["j2d"] = trans_proj_j2d(out, ex["Ks"], root_xyz=root)
        d = lf(e2, out, args.losses, "FreiHand", args)
        loss = sum(d[k] for k in args.losses)
        if case == "fwd_loss":
            return loss.detach()
        if case == "fwd_loss_bwd_flat":
            flat.zero_grad()
        else:
            for p in model.parameters(): p.grad = None
        loss.backward()
        return loss.detach()
    o = run_capture(f); print("ok", float(o))
elif case == "full_nograd":
    from hifihr_amd.models import Model
    from hifihr_amd import options, synth
    from hifihr_amd.traineval import data_dic
    args = options.baseline_config2_args(train_batch=B)
    model = Model(True, torch.device(dev), False, "mano", False, "res18", mano_tables=t).cuda().train()
    ex = data_dic(synth.make_batch(model.hand_layer.handle, model.renderer_p3d, B, device=dev), "FreiHand", "training", args, device=dev)
    root = ex["joints"][:, 9, :].unsqueeze(1)
    def f():
        with torch.no_grad():
            return model("FreiHand", True, ex["imgs"], Ks=ex["Ps"], root_xyz=root)["re_img"]
    o = run_capture(f); print("ok", float(o.mean()))
'''
for c in (sys.argv[1:] or CASES):
    r = subprocess.run([sys.executable, "-c", TEMPLATE % (R, R, c)], capture_output=True, text=True)
    tail = (r.stdout.strip().splitlines() or [""])[-1]
    err = [l for l in r.stderr.splitlines() if "Error" in l or "error" in l or "Segmentation" in l]
    print(f"{c:16s} rc={r.returncode:4d} {tail[:80]} {err[-1][:160] if err else ''}")
