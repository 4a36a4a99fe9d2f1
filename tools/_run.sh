cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests/test_gpu_gemm.py tests/test_gpu_conv.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
bash tools/r06_ab_lib.sh tn_store_interleave tools/_probe/libhifihr_prev.so 3
