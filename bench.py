#!/usr/bin/env python3
"""Headline benchmark: training-step images/sec on FreiHAND-shaped 224x224 batches (BASELINE.json configs[1]:
batch 32 per GPU, ResNet-18 encoder + MANO LBS + render + silhouette/texture losses), one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) carrying `roofline` for the dominant hand-written
kernel and, at N=1, `cpu_baseline` (the oracle step timed on the host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch
import torch.distributed as dist


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (BASELINE configs[1]: 32)")
    ap.add_argument("--encoder", default="res18", choices=["res18", "res50", "res101", "effb3"], help="res18 = BASELINE configs[1] (headline)")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3],
                    help="2 = BASELINE configs[1] (headline); 3 = configs[2] full_rhd_freihand.json: effb3, batch 48, texture + "
                         "perceptual losses, MANO + texture stand-in for the unavailable NIMBLE layer (not a headline line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", type=int, default=-1, help="0: eager; -1 or 1: hipGraph replay (N = 1: whole step; N > 1: forward + backward, then all-reduce + Adam); 2: force the N > 1 form")
    ap.add_argument("--cpu-batch", type=int, default=8, help="sample size of the CPU baseline (images)")
    return ap.parse_args()


def cpu_baseline(args_ns, examples, tables, nimg):
    """The oracle training step (oracle/model_oracle.py) on the host cores, on `nimg` images of the same batch."""
    from oracle.model_oracle import OracleModel, oracle_step
    torch.manual_seed(0)
    model = OracleModel(tables).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-6)
    ex = {k: (v[:nimg].detach().cpu() if torch.is_tensor(v) else v) for k, v in examples.items()}
    t0, nstep = time.time(), 0
    while nstep < 4 and time.time() - t0 < 10.0:          # bounded sample: >= 10 s of CPU work, at most 4 steps
        oracle_step(model, ex, args_ns, opt)
        nstep += 1
    dt = time.time() - t0
    return {"value": nimg * nstep / dt, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{nstep} oracle training step(s) (torch-CPU encoder + C/torch oracle MANO/render/losses + Adam; the C "
                      f"rasteriser is single-threaded) on {nimg} images of the same synthetic batch, {dt:.1f} s"}


def data_path_probe(B, dev, args_ns, n_cache=512, reps=20):
    """Batch assembly on the device (hifihr_amd/data.py, csrc/augment.hip: SURVEY.md 8(f) N1): a synthetic uint8 cache resident
    in HBM, one gather-and-warp launch + the K / joint / vertex products per batch, then data_dic.  Reported next to the step
    (the timed step above replays a resident batch); wall time per batch including the host-side coefficient arithmetic."""
    import numpy as np
    from hifihr_amd.data import FreiHandDeviceCache
    from hifihr_amd.traineval import data_dic
    rng = np.random.RandomState(0)
    cache = FreiHandDeviceCache(rng.randint(0, 256, size=(n_cache, 224, 224, 3)).astype(np.uint8),
                                (rng.rand(n_cache, 224, 224) > 0.5).astype(np.uint8) * 255,
                                np.tile(np.array([[500.0, 0, 112], [0, 500.0, 112], [0, 0, 1]], np.float32), (n_cache, 1, 1)),
                                rng.randn(n_cache, 21, 3).astype(np.float32), rng.randn(n_cache, 778, 3).astype(np.float32), device=dev)
    gen = torch.Generator().manual_seed(0)
    idx = torch.randint(0, n_cache, (B,), generator=gen)
    for _ in range(3):
        data_dic(cache.batch(idx, generator=gen), "FreiHand", "training", args_ns, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        data_dic(cache.batch(idx, generator=gen), "FreiHand", "training", args_ns, device=dev)
    torch.cuda.synchronize()
    wall_us = (time.perf_counter() - t0) / reps * 1e6
    coef = torch.zeros(B, 6, dtype=torch.int32, device=dev); coef[:, 0] = 65536; coef[:, 4] = 65536; coef[:, 2] = 32768; coef[:, 5] = 32768
    oi = torch.empty(B, 3, 224, 224, device=dev); om = torch.empty(B, 3, 224, 224, device=dev)
    idx_d = idx.to(torch.int32).to(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        cache.lib.freihand_augment(cache.images, cache.masks, idx_d, coef, oi, om)
    e1.record()
    torch.cuda.synchronize()
    k_us = e0.elapsed_time(e1) * 1e3 / reps
    alg = B * 224 * 224 * (4 + 1 + 2 * 3 * 4)                  # RGBX + mask byte read, two float [3,H,W] tensors written
    return {"wall_us_per_batch": wall_us, "augment_kernel_us": k_us, "augment_algorithmic_bytes": alg,
            "augment_GBps": alg / (k_us * 1e-6) / 1e9, "images_per_sec_wall": B / (wall_us * 1e-6),
            "note": "uint8 dataset cache resident in HBM; gather + nearest-neighbour affine warp + to_tensor on the device, bit-exact with the reference's PIL path"}


def main():
    a = parse()
    from hifihr_amd import dist as hdist, ops, options, synth
    from hifihr_amd.losses import LossFunction
    from hifihr_amd.mano_tables import synthetic_mano_tables
    from hifihr_amd.models import Model
    from hifihr_amd.optim import FlatParams, FusedAdam
    from hifihr_amd.traineval import data_dic, train_step

    rank, local_rank, world = hdist.init_process_group_from_env()
    assert world == a.gpus or world == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP hot path has no CPU fallback)"
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    # Work on a non-default stream from the start: autograd pins each parameter's gradient accumulation to the stream
    # of its first use, and a step that ever ran on the legacy default stream cannot be captured into a hipGraph later
    # (capture_end crashed / replays produced NaN in round 1: DESIGN.md section 6).
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))

    if a.config == 3:
        a.encoder, a.no_cpu_baseline = "effb3", True
        if a.batch == 32:
            a.batch = 48
        args_ns = options.baseline_config3_args(train_batch=a.batch)
    else:
        args_ns = options.baseline_config2_args(train_batch=a.batch)
    tables = synthetic_mano_tables(0)
    torch.manual_seed(0)
    model = Model(ifRender=True, device=dev, if_4c=False, hand_model="mano", use_mean_shape=False, pretrain=a.encoder, texture_stand_in=10 if a.config == 3 else 0,
                  mano_tables=tables).to(dev).train()
    flat = FlatParams(model)
    hdist.broadcast_params(flat)
    reducer = hdist.GradReducer(flat, num_buckets=4)
    lr = args_ns.force_init_lr if args_ns.force_init_lr > 0 else args_ns.init_lr
    opt = FusedAdam(flat, lr=lr, betas=(0.9, 0.999), grad_scale=reducer.grad_scale)
    loss_func = LossFunction()

    # rank r owns samples [r*B, (r+1)*B) of the global batch; inputs are resident in HBM before timing starts
    sample = synth.make_batch(model.hand_layer.handle, model.renderer_p3d, a.batch, first_index=rank * a.batch, device=dev)
    examples = data_dic(sample, "FreiHand", "training", args_ns, device=dev)
    torch.cuda.synchronize()

    def step():
        return train_step(model, loss_func, opt, examples, args_ns, backward_hook=reducer.finish)

    for _ in range(a.warmup):
        step()
    # per-kernel HIP-event timing (eager, a few steps, outside the headline timing)
    ops.PROFILE.enable()
    for _ in range(min(5, a.steps)):
        step()
    ops.PROFILE.disable()
    kern = ops.PROFILE.summary()                       # {name: (avg_us, launches)}; eager brackets include launch latency
    # Roofline kernel (render_fwd): HIP events around 20 BACK-TO-BACK launches on this batch's own meshes, on the stream the
    # kernel is launched on.  (The timed region below is one hipGraph launch per step and cannot be bracketed per kernel; an
    # eager bracket around a single launch also counts the host's launch latency.)
    render_us = None
    if ops.PROFILE.last_render is not None:
        h_r, v_r, c_r, cam_r, lc_r, ld_r = ops.PROFILE.last_render
        Br, Hr, Sr = v_r.shape[0], h_r.H, h_r.H * h_r.aa
        rgba_r = torch.empty(Br, 4, Hr, Hr, device=dev); fid_r = torch.empty(Br, Sr, Sr, dtype=torch.int32, device=dev)
        ws_r = h_r.workspace(Br, dev)
        for _ in range(3):
            h_r.lib.render_fwd(h_r.h, v_r, c_r, cam_r, lc_r, ld_r, rgba_r, fid_r, ws_r)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            h_r.lib.render_fwd(h_r.h, v_r, c_r, cam_r, lc_r, ld_r, rgba_r, fid_r, ws_r)
        e1.record()
        torch.cuda.synchronize()
        render_us = e0.elapsed_time(e1) * 1e3 / 20

    # Roofline of the DOMINANT kernel (conv_igemm_kernel: ~40 % of the step): every distinct (shape, direction) the step
    # launched, timed with HIP events over 10 back-to-back launches on tensors of that shape, weighted by launches per step.
    conv_roof = None
    nprof = max(1, min(5, a.steps))
    if ops.PROFILE.conv_log:
        from collections import Counter
        lib = ops.get_lib()
        counts = Counter(ops.PROFILE.conv_log)
        tot_flop = tot_us = tot_n = tot_bytes = 0.0
        alg_flop = alg_us = 0.0                        # SURVEY 8(d) accounting: direct-convolution FLOPs of the same layers, and the
        #                                                time of everything that computes them (Winograd: transforms + GEMMs)
        for (geom, direction), cnt in counts.items():
            if direction in ("gemm", "gemm-blas"):     # the 16 batched GEMMs of a Winograd convolution (same kernel, batch = 16)
                _, N_, H_, W_, C_, K_ = geom
                T_ = N_ * ((H_ + 1) // 2) * ((W_ + 1) // 2)
                Vs = torch.randn(16 * T_ * C_, device=dev); Us = torch.randn(16 * K_ * C_, device=dev) * 0.05
                Ms = torch.empty(16 * T_ * K_, device=dev)
                if direction == "gemm-blas":           # these GEMMs run on the vendor library (ops._blas_gemm): not part of the
                    #                                    conv_igemm_kernel roofline, but part of the convolution path's time
                    per_step = cnt / nprof
                    xs = torch.randn(N_ * H_ * W_ * C_, device=dev); wsrc = torch.randn(K_ * 9 * C_, device=dev) * 0.05
                    ys = torch.empty(N_ * H_ * W_ * K_, device=dev)
                    full = lambda: ops._wino_conv(lib, xs, wsrc, ys, None, N_, H_, W_, C_, K_, 0, U=Us)
                    for _ in range(3):
                        full()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        full()
                    e1.record()
                    torch.cuda.synchronize()
                    alg_flop += 2.0 * N_ * H_ * W_ * K_ * 9 * C_ * per_step; alg_us += e0.elapsed_time(e1) * 1e3 / 10 * per_step
                    continue
                nbw = lib.wino_gemm_workspace_bytes(N_, H_, W_, C_, K_)
                wsb = ops._CONV_WS.get(dev) if nbw else None
                if nbw and (wsb is None or wsb.numel() * 4 < nbw):
                    wsb = torch.zeros(nbw // 4 + 64, device=dev); ops._CONV_WS[dev] = wsb
                fn = lambda: lib.wino_gemm(Vs, Us, Ms, N_, H_, W_, C_, K_, ws=wsb)
                for _ in range(3):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / 10
                per_step = cnt / nprof
                tot_flop += 2.0 * 16 * T_ * C_ * K_ * per_step; tot_us += us * per_step; tot_n += per_step
                tot_bytes += 4.0 * 16 * (T_ * C_ + K_ * C_ + T_ * K_) * per_step          # V + U read, M written, once each
                xs = torch.randn(N_ * H_ * W_ * C_, device=dev); wsrc = torch.randn(K_ * 9 * C_, device=dev) * 0.05
                ys = torch.empty(N_ * H_ * W_ * K_, device=dev)
                full = lambda: ops._wino_conv(lib, xs, wsrc, ys, None, N_, H_, W_, C_, K_, 0, U=Us)     # U comes from weight_prep in the step
                full()
                e0.record()
                for _ in range(10):
                    full()
                e1.record()
                torch.cuda.synchronize()
                alg_flop += 2.0 * N_ * H_ * W_ * K_ * 9 * C_ * per_step; alg_us += e0.elapsed_time(e1) * 1e3 / 10 * per_step
                continue
            N_, H_, W_, C_, K_, R_, S_, st_, pd_ = geom
            OH_, OW_ = (H_ + 2 * pd_ - R_) // st_ + 1, (W_ + 2 * pd_ - S_) // st_ + 1
            xs = torch.randn(N_, H_, W_, C_, device=dev); wsrc = torch.randn(K_, R_, S_, C_, device=dev) * 0.05
            ys = torch.randn(N_, OH_, OW_, K_, device=dev)
            wsb = ops._conv_ws(lib, dev, geom, direction == "dgrad")
            if direction == "fwd":
                fn = lambda: lib.conv2d_fwd(xs, wsrc, None, ys, N_, H_, W_, C_, K_, R_, S_, st_, pd_, ws=wsb)
            else:
                scr = torch.empty(wsrc.numel(), device=dev)       # the step gets this transpose from its one weight_prep launch
                lib.weight_transpose(wsrc, scr, K_, R_ * S_, C_)
                fn = lambda: lib.conv2d_bwd_data_pre(ys, scr, xs, N_, H_, W_, C_, K_, R_, S_, st_, pd_, ws=wsb)
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 10
            per_step = cnt / nprof
            flop = 2.0 * N_ * OH_ * OW_ * K_ * R_ * S_ * (3 if C_ == 4 else C_)       # the NHWC4 stem has 3 real channels
            tot_flop += flop * per_step; tot_us += us * per_step; tot_n += per_step
            tot_bytes += 4.0 * (N_ * H_ * W_ * C_ + K_ * R_ * S_ * C_ + N_ * OH_ * OW_ * K_) * per_step    # input, weights, output once each
            alg_flop += flop * per_step; alg_us += us * per_step
        if "weight_prep" in kern:                      # the step's one weight re-layout launch serves all of these layers
            alg_us += kern["weight_prep"][0]
        conv_traffic = None
        tfile = os.path.join(REPO, "profiles", "r01_conv_igemm_traffic.json")
        if a.batch == 32 and a.encoder == "res18" and a.config == 2 and os.path.exists(tfile):
            # PMC-measured HBM bytes per launch of this exact workload (separate FETCH_SIZE / WRITE_SIZE passes, tools/conv_traffic.sh)
            conv_traffic = json.load(open(tfile))["traffic_bytes_per_launch"]
        conv_roof = {"bound": "mfma", "achieved": tot_flop / (tot_us * 1e-6) / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                     "frac": tot_flop / (tot_us * 1e-6) / 1e12 / 157.3, "traffic": conv_traffic,
                     "compulsory_bytes_per_launch": tot_bytes / max(tot_n, 1.0),
                     "kernel": "conv_igemm_kernel (all instantiations: direct forward / backward-data convolutions and those batched GEMMs of "
                               "the Winograd F(2x2,3x3) layers that run on it -- the square 256 / 512-channel ones go to the vendor "
                               "library, ops._blas_gemm --, counted with the FLOPs they actually execute)",
                     "launches_per_step": tot_n, "avg_us": tot_us / max(tot_n, 1.0), "us_per_step": tot_us,
                     "executed_flop_per_step": tot_flop,
                     "algorithmic": {"flop_per_step": alg_flop, "us_per_step": alg_us, "achieved": alg_flop / (alg_us * 1e-6) / 1e12,
                                     "frac": alg_flop / (alg_us * 1e-6) / 1e12 / 157.3,
                                     "note": "SURVEY 8(d) accounting for the same layers: direct-convolution FLOPs (2 N OH OW K R S C) over the "
                                             "time of everything that computes them -- the step's weight re-layout launch and, for the Winograd "
                                             "layers, the input / output transform kernels plus the 16 GEMMs.  `achieved` above is the conservative figure: FLOPs the "
                                             "MFMA kernel actually executes over its own time"},
                     "timing": "HIP events over 10 back-to-back launches of every distinct (shape, direction) of the step, weighted by "
                               "launches per step (weights pre-transposed, as in the step)"}

    use_graph = a.graph != 0
    split = world > 1 or a.graph == 2        # data parallel: graph = forward + backward, then all-reduce + Adam eagerly
    graph_note = "eager"
    gstep, eager_step = None, step
    if use_graph:
        try:
            from hifihr_amd.traineval import GraphedTrainStep
            gstep = GraphedTrainStep(model, loss_func, opt, examples, args_ns, reducer=reducer if split else None)
            step = gstep
            graph_note = ("hipGraph replay of forward + backward, then bucketed all-reduce + fused Adam" if split
                          else "hipGraph replay (whole step captured)")
        except Exception as e:                          # capture is an optimisation; never fail the bench on it
            graph_note = f"eager (hipGraph capture failed: {type(e).__name__}: {str(e)[:200]})"
            opt.graph_mode = False
            reducer.pause_hooks(False)
            torch.cuda.synchronize()
            gstep, step = None, eager_step
        if world > 1:
            # every rank must take the same form of the step (the two forms issue their bucket all-reduces in different orders)
            ok = torch.tensor([1.0 if gstep is not None else 0.0], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) == 0.0 and gstep is not None:
                graph_note = "eager (hipGraph capture failed on another rank)"
                reducer.pause_hooks(False)
                gstep, step = None, eager_step
        if world == 1 and gstep is not None and a.graph == -1:
            # One GPU: the whole-step hipGraph and the eager step (whose weight gradients run on a side stream, ops._AsyncWgrad)
            # are within ~1 % of each other (A/B in one process, tools/time_async_wgrad.py: 7.65 vs 7.74 ms); keep whichever a short
            # trial finds faster on this host.
            def timed1(fn, n=15):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                t_start = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t_start) / n
            t_graph, = (timed1(gstep),)
            opt.graph_mode = False
            t_eager = timed1(eager_step)
            if t_eager < 0.995 * t_graph:
                step = eager_step
                graph_note = (f"eager step, weight gradients on a side stream (chosen over the whole-step hipGraph: "
                              f"{t_eager * 1e3:.2f} vs {t_graph * 1e3:.2f} ms/step in a 15-step trial)")
            else:
                opt.graph_mode = True
                graph_note += f" (chosen over the eager step: {t_graph * 1e3:.2f} vs {t_eager * 1e3:.2f} ms/step in a 15-step trial)"
        if world > 1 and gstep is not None and a.graph == -1:
            # Data parallel has two forms of the step: the hipGraph replay followed by the (not overlapped) bucketed all-reduce, and
            # the eager step whose all-reduce buckets overlap the rest of backward.  On one GPU they run within 0.5 % of each other
            # (the step is GPU-bound, the host runs ahead), so which one wins at N > 1 depends on the exchange time and on the host:
            # measure both for a few steps (max over ranks) and keep the faster -- every rank takes the same decision.
            def timed(fn, n=3):
                fn()
                torch.cuda.synchronize(); dist.barrier()
                t_start = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize(); dist.barrier()
                t = torch.tensor([time.perf_counter() - t_start], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return float(t.item()) / n
            t_graph = timed(gstep)
            reducer.pause_hooks(False)
            t_eager = timed(eager_step)
            if t_eager < t_graph:
                step = eager_step
                graph_note = (f"eager step, bucketed all-reduce overlapped with backward (chosen over the hipGraph form: "
                              f"{t_eager * 1e3:.2f} vs {t_graph * 1e3:.2f} ms/step in a 3-step trial)")
            else:
                reducer.pause_hooks(True)
                graph_note += f" (chosen over the eager overlapped form: {t_graph * 1e3:.2f} vs {t_eager * 1e3:.2f} ms/step in a 3-step trial)"
        for _ in range(2):
            step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss, loss_dic = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        B = a.batch
        ms = dt / a.steps * 1e3
        out = {
            "metric": "train images/sec, FreiHAND 224x224 (ResNet-18 + MANO LBS + render + losses + Adam)",
            "value": world * B * a.steps / dt, "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (FreiHAND-shaped, seeded; synthetic MANO-shaped tables; random-init weights)",
            "config": {"workload": "BASELINE configs[1]: FreiHAND batch=32/GPU, ResNet-18 encoder + MANO LBS + "
                                   "silhouette/texture render losses, 224x224, aa=3 (672^2 samples)"
                                   + ("" if a.encoder == "res18" else f" [encoder swapped to {a.encoder}: NOT the headline config]")
                                   + ("" if a.config == 2 else " [BASELINE configs[2] composition: full_rhd_freihand.json losses incl. VGG19 "
                                      "perceptual (seeded random weights), MANO + vertex-colour texture stand-in for NIMBLE]"),
                       "per_gpu_batch": B, "global_batch": world * B, "losses": args_ns.losses, "parallelism": f"dp{world}"},
            "loss": float(loss.detach()), "launch_mode": graph_note,
        }
        # roofline of the dominant hand-written kernel: the fused rasterise+shade+resolve forward.
        # algorithmic bytes per image (SURVEY.md 8d / DESIGN.md): verts 778*12 + faces 1538*12 + attrs 778*24 +
        # RGBA 224^2*16 + face-id side buffer 672^2*4
        alg = 778 * 12 + 1538 * 12 + 778 * 24 + 224 * 224 * 16 + 672 * 672 * 4
        if render_us is not None:
            us = render_us
            ach = alg * B / (us * 1e-6) / 1e9
            traffic = None
            tf = os.path.join(REPO, "profiles", "r01_render_fwd_traffic.json")
            if B == 32 and os.path.exists(tf):      # PMC-measured HBM bytes per launch of this exact workload (separate --pmc passes)
                traffic = json.load(open(tf))["traffic_bytes_per_launch"]
            render_roof = {"bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                           "traffic": traffic, "kernel": "render_fwd_kernel<3> (+ render_vertex_kernel)",
                           "avg_us": us, "algorithmic_bytes_per_launch": alg * B,
                           "timing": "HIP events over 20 back-to-back launches on this batch's meshes (kernel + its 8 us vertex pass)"}
            # `roofline` = the dominant kernel of the step (the MFMA convolution); the rasteriser the north star asks an HBM
            # figure for is reported next to it
            out["roofline"] = conv_roof if conv_roof is not None else render_roof
            out["roofline_render_fwd"] = render_roof
        elif conv_roof is not None:
            out["roofline"] = conv_roof
        out["kernels_avg_us_eager"] = {k: round(v[0], 2) for k, v in kern.items()}     # single-launch brackets, incl. launch latency
        rf = render_us if render_us is not None else kern.get("render_fwd", (0,))[0]
        out["render_ms_per_frame"] = {"fwd": rf / B / 1e3, "fwd+bwd": (rf + kern.get("render_bwd", (0,))[0]) / B / 1e3}
        if world == 1:
            out["data_path"] = data_path_probe(B, dev, args_ns)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args_ns, examples, tables, a.cpu_batch)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
