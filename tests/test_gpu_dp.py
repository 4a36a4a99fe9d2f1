"""Data parallelism on the GPU box without a second GPU: two ranks share cuda:0 and exchange over gloo (tools/dp_sync_check.py under
torch.distributed.run).  Checks, with the real model and kernels: the exchanged gradient == the sum of the per-rank gradients and ==
what one process computes for the concatenated batch with per-replica batch-norm statistics; replicas stay bit-identical after
several steps in BOTH step forms (eager with bucketed hooks; hipGraph replay + all-reduce + Adam) and when switching between them."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_stay_in_sync():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HIFIHR_DIST_BACKEND="gloo", DP_CHECK_BATCH="8", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tools", "dp_sync_check.py")],
                       env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:]); print(r.stderr[-3000:])
    assert r.returncode == 0 and "dp sync check ok" in r.stdout


def test_bench_multi_rank_path_runs_end_to_end():
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per GPU) -- here two ranks on the one GPU over gloo,
    a functional run only: every rank must issue the same collectives in the same order through the profiled steps, the three step
    forms' constructors (collective-free) with their agreement all-reduces, the timing trial and the timed region.  Round 3 found the
    in-step kernel profile running its extra steps (which hold the gradient exchange) on rank 0 only: a collective mismatch that no
    single-process run can see."""
    import json
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HIFIHR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--batch", "8"], env=env, capture_output=True, text=True, timeout=900)
    print(r.stdout[-2000:]); print(r.stderr[-3000:])
    assert r.returncode == 0
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16 and d["scaling"] == "weak" and d["value"] > 0
    assert d["rccl"]["world_size"] == 2 and len(d["rccl"]["allreduce_us_per_bucket"]) == d["rccl"]["buckets"]
    assert "trial" in d["launch_mode"]


def test_bench_gpus_2_plain_invocation_starts_two_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher (the form the driver uses at N = 1): bench.py must start the two ranks itself -- fresh
    child processes, the parent never touching the GPU -- and print ONE line with n_gpus == 2.  (Round 3's bench silently ran one rank.)"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HIFIHR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--batch", "8"], env=env, capture_output=True, text=True, timeout=900)
    print(r.stdout[-2000:]); print(r.stderr[-3000:])
    assert r.returncode == 0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp2"
    assert d["rccl"]["world_size"] == 2 and "bench.py itself" in d["rccl"]["launched_by"]


def test_bench_gpus_4_four_ranks_on_one_gpu():
    """`python bench.py --gpus 4 --batch 2`: FOUR ranks (gloo, one GPU) so that the first 8-GPU run is not the first run with more than two
    ranks -- bucket boundaries, the agreement all-reduces and the timing trial with an even world size > 2."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HIFIHR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--batch", "2"], env=env, capture_output=True, text=True, timeout=1200)
    print(r.stdout[-2000:]); print(r.stderr[-3000:])
    assert r.returncode == 0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["config"]["global_batch"] == 8 and d["config"]["parallelism"] == "dp4" and d["scaling"] == "weak"
    assert d["rccl"]["world_size"] == 4 and d["rccl"]["buckets"] == 4 and len(d["rccl"]["allreduce_us_per_bucket"]) == 4


def test_one_sided_capture_failure_falls_back_on_every_rank():
    """Rank 1's hipGraph capture fails (injected), rank 0's succeeds: the constructors are collective-free, so both meet in the agreement
    all-reduce and BOTH run the eager step -- no mismatched collective, the bench line says why."""
    import json
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HIFIHR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", HIFIHR_TEST_FAIL_CAPTURE_RANK="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--batch", "8"], env=env, capture_output=True, text=True, timeout=900)
    print(r.stdout[-1500:]); print(r.stderr[-3000:])
    assert r.returncode == 0
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["launch_mode"].startswith("eager") and "another rank" in d["launch_mode"]


@pytest.mark.parametrize("dataset", ["FreiHand", "HO3D"])
def test_train_front_end_multi_rank(tmp_path, dataset):
    """train_hrnet.py under torch.distributed.run with two ranks (one GPU, gloo): graph capture agreed between the ranks, the captured
    step + bucket exchange + Adam, periodic test / challenge dump and checkpoint on rank 0, clean shutdown."""
    import json
    cfg = json.load(open(os.path.join(ROOT, "tests", "data", "nimble_style_config.json")))
    cfg.update(base_out_path=str(tmp_path / "run"), train_batch=4, val_batch=8, total_epochs=1, pretrain="res18", hand_model="mano",
               losses=["joint_3d", "mpose", "mshape", "texture", "mrgb", "sil", "ssim_tex"])
    f = tmp_path / "cfg.json"
    f.write_text(json.dumps(cfg))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HIFIHR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "train_hrnet.py"), "--config_json", str(f), "--synthetic_size", "32", "--print_freq", "2"]
    if dataset == "HO3D":
        cmd += ["--dataset", "HO3D"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    print(r.stdout[-3000:]); print(r.stderr[-3000:])
    assert r.returncode == 0 and "Done!" in r.stdout and "nan" not in r.stdout.lower()
    assert (tmp_path / "run" / "model" / "texturehand_latest.t7").exists()


def test_rccl_wrapper_single_rank_roundtrip():
    """csrc/comm.hip through the C ABI: a world-size-1 communicator on this GPU; the SUM all-reduce and the broadcast leave the buffer
    unchanged (the multi-rank path is the same calls; 8-GPU runs are the driver's)."""
    import torch
    from hifihr_amd._lib import get_lib
    lib = get_lib()
    uid = lib.comm_unique_id()
    assert len(uid) == 128
    h = lib.comm_init(0, 1, uid)
    try:
        buf = torch.randn(1 << 20, device="cuda")
        ref = buf.clone()
        lib.comm_allreduce(h, buf)
        lib.comm_allreduce(h, buf[1000:5000])          # a bucket = a slice of the flat buffer
        lib.comm_broadcast(h, buf, 0)
        torch.cuda.synchronize()
        assert torch.equal(buf, ref)
    finally:
        lib.comm_destroy(h)
