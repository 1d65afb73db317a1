cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/ab/lstm_time.py packed 2>&1 | grep -v amdgpu.ids
timeout -k 10 900 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_fusion_multitile.py tests/test_gpu_f16_range.py tests/test_gpu_backward.py -q -m gpu -x 2>&1 | tail -4
