#!/usr/bin/env python3
"""Training / evaluation front-end with the reference's command line:  python train_hrnet.py --config_json <file>.json

Mirrors the flow of the reference's train_hrnet.py (argument overlay :503-519, model / optimizer / scheduler / checkpoint
set-up :537-566, epoch driver :449-485 with the lambda schedules, periodic test + save, evaluation mode :487-496) on this
package's device-resident pieces:

  data        hifihr_amd.data.FreiHandDeviceCache -- the decoded set lives in HBM, a batch is one gather-and-warp launch.
              Source: --freihand_cache <npz with images u8 [n,224,224,3], masks u8, Ks, joints, verts> (pre-decoded by the user;
              JPEG decoding is outside the hot path) or, by default, a seeded synthetic FreiHAND-shaped set (--synthetic_size).
              --dataset HO3D: hifihr_amd.data.HO3DDeviceCache -- 480 x 640 frames in HBM, the reference's hand crop (window from the
              projected joints, Pillow-exact crop + resize to 224) on the device; --ho3d_cache <npz> or seeded synthetic frames.  The epoch
              driver's periodic test and the evaluation mode write the challenge dump `<base_out_path>/json/test/<epoch>/pred.json` from the
              evaluation split (--ho3d_eval_cache <npz>: hand boxes + root joints; reference train_hrnet.py:124-136, 286-293).
  step        hifihr_amd.traineval.GraphedTrainStep (hipGraph replay) or the eager step (--graph 0)
  multi-GPU   one process per GPU under torch.distributed.run; rank r takes every world-th batch slice; RCCL all-reduce of the
              flat gradient buffer (hifihr_amd/dist.py).  Replaces nn.DataParallel (:560).
  checkpoints the reference's .t7 layout (hifihr_amd/checkpoint.py); --pretrain_model resumes a reference file.

hand_model "nimble" configs run with MANO + the texture stand-in (hifihr_amd/models.py) and say so: the NIMBLE submodule
and its assets are not part of the reference tree.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch


def parse(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--config_json", default=None)
    ap.add_argument("--freihand_cache", default=None, help="npz: images, masks, Ks, joints, verts [, eval_* counterparts]")
    ap.add_argument("--dataset", default="FreiHand", choices=["FreiHand", "HO3D"],
                    help="HO3D: train on 480 x 640 frames through hifihr_amd.data.HO3DDeviceCache (the reference's hand crop on the device)")
    ap.add_argument("--ho3d_eval_cache", default=None, help="npz of the HO-3D EVALUATION split: images u8 [n,480,640,3], Ks [n,3,3], bboxes "
                    "[n,2,2] ((x0, y0), (x1, y1)), root_xyz [n,3] (the split has a hand box and the root joint, no 21 joints: dataset.py:1071-1080); "
                    "without it the evaluation pass runs on the training frames with boxes / roots derived from their joints")
    ap.add_argument("--ho3d_cache", default=None, help="npz: images u8 [n,480,640,3], hand_masks u8 [n,480,640], Ks [n,3,3] (camMat . cam_extr), "
                                                       "xyz21 [n,21,3]; default: seeded synthetic frames")
    ap.add_argument("--synthetic_size", type=int, default=512)
    ap.add_argument("--mano_pkl", default=None, help="MANO_RIGHT.pkl (licensed, user supplied); default: synthetic MANO-shaped tables")
    ap.add_argument("--nimble_layer", choices=["mano-stand-in", "synthetic", "synthetic-uv"], default="mano-stand-in",
                    help="what hand_model 'nimble' runs on: MANO + the vertex-colour texture stand-in, or the NIMBLE-shaped layer "
                         "(csrc/lbs.hip: 25 joints, 5990 skin vertices, texture PCA) on seeded synthetic tables; synthetic-uv: the same "
                         "layer with a texture IMAGE sampled through per-face uvs (TexturesUV, hifihr_render_fwd_uv)")
    ap.add_argument("--graph", type=int, default=1)
    ap.add_argument("--max_iters", type=int, default=0, help="stop after this many iterations (0 = run the epochs)")
    ap.add_argument("--print_freq", type=int, default=50)
    ap.add_argument("--dist_timeout_min", type=float, default=60.0, help="collective timeout of the process group: has to cover rank 0's "
                    "evaluation pass, during which the other ranks wait in a barrier")
    ap.add_argument("--override", default=None, help='JSON object of option overrides, e.g. \'{"total_epochs": 2}\'')
    return ap.parse_args(argv)


def build_args(cli):
    from hifihr_amd import options
    over = json.loads(cli.override) if cli.override else {}
    defaults = dict(mode=["training"], save_interval=10, if_test=True, save_mode="only_latest", pretrain_model=None,
                    base_out_path="outputs/run", val_batch=16)
    defaults.update(over)
    args = options.make_args(cli.config_json, **{k: v for k, v in defaults.items() if cli.config_json is None or k in over})
    for k, v in defaults.items():                      # keys a JSON may leave out
        if not hasattr(args, k):
            setattr(args, k, v)
    args.state_output = os.path.join(args.base_out_path, "model")       # options/train_options.py:208-220
    args.pred_output = os.path.join(args.base_out_path, "json")         # :214 (the pred.json dumps)
    args.texture_stand_in = 0
    if args.hand_model == "nimble" and cli.nimble_layer in ("synthetic", "synthetic-uv"):
        print("[train_hrnet] hand_model 'nimble': NIMBLE-shaped layer on seeded synthetic tables (hifihr_amd/nimble_tables.py); the real "
              "NIMBLE assets are not available (SURVEY.md section 8 A9)")
    elif args.hand_model == "nimble":
        print("[train_hrnet] hand_model 'nimble': the NIMBLE layer is not available; running MANO + the 10-component "
              "vertex-colour texture stand-in (SURVEY.md section 8 A9)")
        args.hand_model, args.texture_stand_in = "mano", 10
    if isinstance(args.mode, str):
        args.mode = [args.mode]
    return args


def load_or_make_dataset(cli, model, device):
    """-> (train arrays, eval arrays) as dicts of numpy arrays: images u8 [n,H,W,3], masks u8 [n,H,W], Ks, joints, verts."""
    if cli.freihand_cache:
        z = np.load(cli.freihand_cache)
        tr = {k: z[k] for k in ("images", "masks", "Ks", "joints", "verts")}
        ev = {k: z["eval_" + k] for k in tr} if "eval_images" in z else None
        return tr, ev
    from hifihr_amd import synth
    parts, n = [], cli.synthetic_size + 64
    if model.hand_model == "nimble":                  # the data side is MANO either way (FreiHAND's ground truth)
        from hifihr_amd import ops
        from hifihr_amd.mano_tables import synthetic_mano_tables
        mt = synthetic_mano_tables(0)
        mano, rend = ops.ManoLayerHandle(mt), ops.RendererHandle(mt.faces, 778, image_size=224, aa=3)
    else:
        mano, rend = model.hand_layer.handle, model.renderer_p3d
    for first in range(0, n, 64):
        s = synth.make_batch(mano, rend, min(64, n - first), first_index=first, device=device, images="render")
        parts.append({"images": (s["trans_images"].permute(0, 2, 3, 1) * 255).round().to(torch.uint8).numpy(),
                      "masks": (s["trans_masks"][:, 0] * 255).to(torch.uint8).numpy(), "Ks": s["trans_Ks"].numpy(),
                      "joints": s["trans_joints"].numpy(), "verts": s["trans_verts"].numpy()})
    full = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
    cut = cli.synthetic_size
    return {k: v[:cut] for k, v in full.items()}, {k: v[cut:] for k, v in full.items()}


def train_ho3d(cli, args, model, loss_func, opt, sched, reducer, current_epoch, rank, world, device, say):
    """The epoch driver on HO-3D frames (reference train_hrnet.py:449-485 with dat_name 'HO3D'): batches from HO3DDeviceCache through the
    HO3D branch of data_dic, copied into the captured step's static inputs."""
    from hifihr_amd import options, synth
    from hifihr_amd.checkpoint import save_model
    from hifihr_amd.data import HO3DDeviceCache
    from hifihr_amd.traineval import GraphedTrainStep, data_dic, train_step
    if cli.ho3d_cache:
        z = np.load(cli.ho3d_cache)
        frames = {"images_u8": z["images"], "hand_masks_u8": z["hand_masks"], "Ks": z["Ks"], "xyz21": z["xyz21"]}
    else:
        from hifihr_amd import ops
        from hifihr_amd.mano_tables import synthetic_mano_tables
        mt = synthetic_mano_tables(0)
        frames = synth.make_ho3d_frames(ops.ManoLayerHandle(mt), ops.RendererHandle(mt.faces, 778, image_size=224, aa=3), cli.synthetic_size,
                                        device=device, images="render")
    cache = HO3DDeviceCache(**frames, device=device)
    say(f"[train_hrnet] HO3D: {cache.n} frames resident on {device}; world {world}; encoder {args.pretrain}; losses {args.losses}")
    eval_cache = ho3d_eval_cache(cli, frames, device) if rank == 0 else None
    if "training" not in args.mode:                       # evaluation only (reference train_hrnet.py:487-496)
        if rank == 0:
            say("[train_hrnet] HO3D evaluation:", run_evaluation_ho3d(model, eval_cache, args, device, current_epoch))
        if world > 1:                                      # the other ranks leave together with rank 0, not while it still evaluates
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return 0
    B = args.train_batch
    gen = torch.Generator().manual_seed(1000 + current_epoch)
    noise_gen = torch.Generator().manual_seed(77 + rank)
    stepper, stepper_key, it, t_last = None, None, 0, time.perf_counter()
    for epoch in range(1, args.total_epochs + 1 - current_epoch):
        options.update_lambdas_for_epoch(args, epoch + current_epoch)
        lam_key = (args.lambda_pose, args.lambda_j2d_gt, args.lambda_shape, args.lambda_tex_reg)
        if stepper is not None and lam_key != stepper_key:
            stepper.release()
            stepper = None
        perm = torch.randperm(cache.n, generator=gen)
        per_step = B * world
        for lo in range(0, cache.n - per_step + 1, per_step):
            idx = perm[lo + rank * B: lo + (rank + 1) * B]
            ex = data_dic(cache.batch(idx, generator=noise_gen), "HO3D", "training", args, device=device)
            if cli.graph and stepper is None:
                ok = 1
                try:
                    stepper = GraphedTrainStep(model, loss_func, opt, ex, args, dat_name="HO3D", reducer=reducer if world > 1 else None)
                    stepper_key = lam_key
                except Exception as e:            # noqa: BLE001
                    # GraphedTrainStep's warm-up and capture are collective-free (round 3; the round-2 advisor found the bucket all-reduces
                    # of the old warm-up left the other ranks inside collectives a failing rank never joined), so a one-sided failure
                    # meets the other ranks in the flag all-reduce below and every rank falls back to the eager step together
                    print(f"[train_hrnet] rank {rank}: hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr, flush=True)
                    ok = 0
                if world > 1:
                    flag = torch.tensor([ok], device=device, dtype=torch.int32)
                    torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
                    ok = int(flag.item())
                if not ok:
                    if stepper is not None:
                        stepper.release()
                    stepper, cli.graph = None, 0
            if stepper is not None:
                stepper.load_batch(ex)
                loss, dic = stepper()
            else:
                loss, dic = train_step(model, loss_func, opt, ex, args, dat_name="HO3D", backward_hook=reducer.finish)
            it += 1
            if it % cli.print_freq == 0 or it == cli.max_iters:
                torch.cuda.synchronize()
                dt, t_last = time.perf_counter() - t_last, time.perf_counter()
                n_it = cli.print_freq if it % cli.print_freq == 0 else it % cli.print_freq
                terms = " ".join(f"{k}={float(dic[k].detach()):.4g}" for k in args.losses)
                say(f"[train_hrnet] epoch {epoch + current_epoch} it {it} loss {float(loss.detach()):.5f} ({terms}) {n_it * per_step / dt:.0f} img/s")
            if cli.max_iters and it >= cli.max_iters:
                break
        if (epoch + current_epoch) % args.save_interval == 0 or (cli.max_iters and it >= cli.max_iters):
            if rank == 0:
                say("[train_hrnet] saved", save_model(model, opt, sched, epoch, current_epoch, args))
                # the periodic test of the reference's epoch driver (:470-480): on HO-3D it is the challenge dump
                say("[train_hrnet] HO3D test:", run_evaluation_ho3d(model, eval_cache, args, device, epoch + current_epoch))
            if world > 1:
                # rank 0 alone walks the evaluation split: the others wait HERE, so that the evaluation is not booked as step time of the
                # next epoch.  This barrier is a collective like any other: it runs under the process group's timeout (`--dist_timeout_min`,
                # set at init_process_group), which therefore has to cover the longest evaluation pass.
                torch.distributed.barrier()
        sched.step()
        if cli.max_iters and it >= cli.max_iters:
            break
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    say("Done!")
    return 0


def ho3d_eval_cache(cli, frames, device):
    """The HO-3D evaluation split as an HO3DDeviceCache (crop window from the hand box, root joint instead of 21 joints: reference
    data/dataset.py:1071-1080).  From --ho3d_eval_cache, else derived from the training frames: box = the projected joints' bounds,
    root = joint 0 of the HO-3D order."""
    from hifihr_amd.data import HO3DDeviceCache
    if cli.ho3d_eval_cache:
        z = np.load(cli.ho3d_eval_cache)
        n = z["images"].shape[0]
        masks = z["hand_masks"] if "hand_masks" in z else np.zeros((n,) + z["images"].shape[1:3], np.uint8)
        return HO3DDeviceCache(z["images"], masks, z["Ks"], np.zeros((n, 21, 3), np.float32) + z["root_xyz"][:, None], device=device,
                               bboxes=z["bboxes"], root_xyz=z["root_xyz"])
    Ks, xyz = np.asarray(frames["Ks"], np.float32), np.asarray(frames["xyz21"], np.float32)
    uvw = np.einsum("nij,nkj->nki", Ks, xyz)
    uv = uvw[..., :2] / uvw[..., 2:3]
    boxes = np.stack([uv.min(1), uv.max(1)], 1)
    return HO3DDeviceCache(frames["images_u8"], frames["hand_masks_u8"], Ks, xyz, device=device, bboxes=boxes, root_xyz=xyz[:, 0])


def run_evaluation_ho3d(model, cache, args, device, epoch):
    """The reference's HO-3D evaluation pass (train_hrnet.py:55-64 the evaluation queries, :124-136 joints back in the HO-3D order and
    OpenGL axes, :286-293 `pred.json` for the challenge server; texture metrics when rendering): no ground truth exists for this split,
    the product is the dump.  -> (path, n, texture metrics)."""
    from hifihr_amd.evaluate import Evaluator
    from hifihr_amd.traineval import data_dic
    ev = Evaluator()
    model.eval()
    with torch.no_grad():
        for lo in range(0, cache.n, args.val_batch):
            idx = torch.arange(lo, min(cache.n, lo + args.val_batch))
            zeros = np.zeros((len(idx), 2), np.float32)
            ex = data_dic(cache.batch(idx, center_noise=zeros, scale_noise=np.ones(len(idx), np.float32)), "HO3D", "evaluation", args, device=device)
            out = model("HO3D", False, ex["imgs"], Ks=ex["Ps"], root_xyz=ex["root_xyz"].unsqueeze(1))
            ev.collect(out, ex, "HO3D", render=args.render)
    model.train()
    path = os.path.join(args.pred_output, "test", str(epoch), "pred.json")                     # train_hrnet.py:286-290
    n, _ = ev.dump(path)
    return path, n, ev.summary()


def run_evaluation(model, cache, arrays, args, device):
    from hifihr_amd.evaluate import Evaluator
    from hifihr_amd.traineval import data_dic
    ev = Evaluator()
    model.eval()
    n = cache.n
    with torch.no_grad():
        for lo in range(0, n, args.val_batch):
            idx = torch.arange(lo, min(n, lo + args.val_batch))
            ex = data_dic(cache.batch(idx, rots=np.zeros(len(idx))), "FreiHand", "training", args, device=device)
            root = ex["joints"][:, args.ROOT, :].unsqueeze(1)
            out = model("FreiHand", False, ex["imgs"], Ks=ex["Ps"], root_xyz=root)
            ev.collect(out, ex, "FreiHand", render=args.render)
    model.train()
    return ev.summary(arrays["joints"], arrays["verts"])


def main(argv=None):
    cli = parse(argv)
    args = build_args(cli)
    from hifihr_amd import dist as hdist
    from hifihr_amd import options
    from hifihr_amd.checkpoint import freeze_model_modules, load_model, save_model
    from hifihr_amd.data import FreiHandDeviceCache
    from hifihr_amd.losses import LossFunction
    from hifihr_amd.mano_tables import load_mano_pkl, synthetic_mano_tables
    from hifihr_amd.models import Model
    from hifihr_amd.optim import FlatParams, FusedAdam
    from hifihr_amd.traineval import GraphedTrainStep, data_dic, train_step

    rank, local_rank, world = hdist.init_process_group_from_env(timeout_min=cli.dist_timeout_min)
    assert torch.cuda.is_available(), "the hot path has no CPU fallback"
    device = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)
    torch.cuda.set_stream(torch.cuda.Stream(device=device))            # see GraphedTrainStep: never step on the legacy default stream
    args.device = device
    say = print if rank == 0 else (lambda *a, **k: None)

    tables = load_mano_pkl(cli.mano_pkl) if cli.mano_pkl else synthetic_mano_tables(0)
    torch.manual_seed(0)
    nimble_tables = None
    if args.hand_model == "nimble" and cli.nimble_layer == "synthetic-uv":
        from hifihr_amd.nimble_tables import add_synthetic_uv, synthetic_nimble_tables
        nimble_tables = add_synthetic_uv(synthetic_nimble_tables(0))
    model = Model(ifRender=args.render, device=device, if_4c=args.four_channel, hand_model=args.hand_model,
                  use_mean_shape=args.use_mean_shape, pretrain=args.pretrain, root_id=args.ROOT, root_id_nimble=args.ROOT_NIMBLE,
                  ifLight=args.light_estimation, mano_tables=tables, texture_stand_in=args.texture_stand_in,
                  nimble_tables=nimble_tables).to(device).train()
    frozen = freeze_model_modules(model, args)          # only_train_regressor / only_train_texture (train_hrnet.py:566)
    if frozen:
        say("[train_hrnet] frozen:", ", ".join(frozen))
    flat = FlatParams(model)
    hdist.broadcast_params(flat)
    reducer = hdist.GradReducer(flat, num_buckets=4)
    wd = 0.01 if args.optimizer == "AdamW" else 0.0                   # train_hrnet.py:549-550: "AdamW" is Adam with L2 0.01
    opt = FusedAdam(flat, lr=args.init_lr, betas=(0.9, 0.999), weight_decay=wd, grad_scale=reducer.grad_scale)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=args.lr_steps, gamma=args.lr_gamma)
    model, current_epoch, opt, sched = load_model(model, opt, sched, args)
    if args.force_init_lr > 0:
        opt.param_groups[0]["lr"] = args.force_init_lr
    loss_func = LossFunction()

    dat_name = cli.dataset
    if dat_name == "HO3D":
        return train_ho3d(cli, args, model, loss_func, opt, sched, reducer, current_epoch, rank, world, device, say)
    train_arrays, eval_arrays = load_or_make_dataset(cli, model, device)
    cache = FreiHandDeviceCache(train_arrays["images"], train_arrays["masks"], train_arrays["Ks"], train_arrays["joints"],
                                train_arrays["verts"], device=device)
    eval_cache = None
    if eval_arrays is not None and len(eval_arrays["images"]):
        eval_cache = FreiHandDeviceCache(eval_arrays["images"], eval_arrays["masks"], eval_arrays["Ks"], eval_arrays["joints"],
                                         eval_arrays["verts"], device=device)
    say(f"[train_hrnet] {cache.n} training samples resident on {device}; world {world}; encoder {args.pretrain}; losses {args.losses}")

    if "evaluation" in args.mode:
        say("[train_hrnet] evaluation:", run_evaluation(model, eval_cache or cache, eval_arrays or train_arrays, args, device))
        return 0

    B = args.train_batch
    gen = torch.Generator().manual_seed(1000 + current_epoch)         # same permutation on every rank
    rot_gen = torch.Generator().manual_seed(77 + rank)                # ... different rotations
    stepper, stepper_key, it, t_last = None, None, 0, time.perf_counter()
    for epoch in range(1, args.total_epochs + 1 - current_epoch):
        options.update_lambdas_for_epoch(args, epoch + current_epoch)
        lam_key = (args.lambda_pose, args.lambda_j2d_gt, args.lambda_shape, args.lambda_tex_reg)
        if stepper is not None and lam_key != stepper_key:
            stepper.release()
            stepper = None                        # the loss weights are kernel arguments baked into the captured graph: re-capture
        perm = torch.randperm(cache.n, generator=gen)
        per_step = B * world
        for lo in range(0, cache.n - per_step + 1, per_step):
            idx = perm[lo + rank * B: lo + (rank + 1) * B]
            # the batch is written straight into the captured step's static inputs (one staged copy + two launches)
            ex = cache.batch_examples(idx, generator=rot_gen, out=stepper.static if stepper is not None else None, root_id=args.ROOT)
            if cli.graph and stepper is None:
                ok = 1
                try:
                    stepper = GraphedTrainStep(model, loss_func, opt, ex, args, reducer=reducer if world > 1 else None)
                    stepper_key = lam_key
                except Exception as e:            # noqa: BLE001  -- report and continue eagerly (GraphedTrainStep restored the state)
                    # (collective-free constructor: the ranks meet in the flag all-reduce below -- see the HO-3D loop above)
                    print(f"[train_hrnet] rank {rank}: hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr, flush=True)
                    ok = 0
                if world > 1:                     # the two step forms issue their bucket all-reduces in different orders: all ranks
                    flag = torch.tensor([ok], device=device, dtype=torch.int32)       # must take the same one
                    torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
                    ok = int(flag.item())
                if not ok:
                    if stepper is not None:
                        stepper.release()
                    stepper, cli.graph = None, 0
            if stepper is not None:
                if ex["imgs"].data_ptr() != stepper.static["imgs"].data_ptr():      # the batch the step was captured on
                    stepper.load_batch(ex)
                loss, dic = stepper()
            else:
                loss, dic = train_step(model, loss_func, opt, ex, args, backward_hook=reducer.finish)
            it += 1
            if it % cli.print_freq == 0 or it == cli.max_iters:
                torch.cuda.synchronize()
                dt = time.perf_counter() - t_last
                t_last = time.perf_counter()
                n_it = cli.print_freq if it % cli.print_freq == 0 else it % cli.print_freq
                terms = " ".join(f"{k}={float(dic[k].detach()):.4g}" for k in args.losses)
                say(f"[train_hrnet] epoch {epoch + current_epoch} it {it} loss {float(loss.detach()):.5f} ({terms}) {n_it * per_step / dt:.0f} img/s")
            if cli.max_iters and it >= cli.max_iters:
                break
        if (epoch + current_epoch) % args.save_interval == 0 or (cli.max_iters and it >= cli.max_iters):
            if args.if_test and eval_cache is not None:
                say("[train_hrnet] test:", run_evaluation(model, eval_cache, eval_arrays, args, device))
            if rank == 0:
                say("[train_hrnet] saved", save_model(model, opt, sched, epoch, current_epoch, args))
        sched.step()
        if cli.max_iters and it >= cli.max_iters:
            break
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    say("Done!")
    return 0


if __name__ == "__main__":
    sys.exit(main())
