cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_gpu_f16_range.py tests/test_gpu_dense.py tests/test_gpu_backward.py tests/test_gpu_fusion_multitile.py -x -q 2>&1 | tail -8
