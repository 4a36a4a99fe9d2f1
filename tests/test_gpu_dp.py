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
