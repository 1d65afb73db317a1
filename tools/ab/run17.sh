cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python tools/ab/lstm_time.py carry 2>&1 | grep lstm
timeout -k 10 900 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_fusion_multitile.py tests/test_gpu_f16_range.py -x -q 2>&1 | tail -5
