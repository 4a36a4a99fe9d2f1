cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests/test_gpu_gemm.py tests/test_gpu_conv.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
timeout 120 python3 tools/gemm_stamp4.py 2>&1 | grep -v amdgpu.ids
bash tools/r06_ab_lib.sh store_interleave tools/_probe/libhifihr_prev.so 3
