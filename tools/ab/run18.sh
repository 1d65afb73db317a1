cd $GRAFT_REPO_ROOT
SAGNN_LIB=$PWD/scratch/ab/base.so python tools/ab/mhsa_time.py base 2>&1 | grep mhsa
python tools/ab/mhsa_time.py wl 2>&1 | grep mhsa
timeout -k 10 900 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_fusion_multitile.py tests/test_gpu_f16_range.py tests/test_gpu_backward.py -x -q 2>&1 | tail -5
