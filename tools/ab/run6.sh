cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/ab/combine_test.py ref dpp_ln 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/ab/dbg_d32.py 2>&1 | grep -v amdgpu.ids | cut -c1-120
timeout -k 10 900 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_fusion_multitile.py tests/test_gpu_f16_range.py tests/test_gpu_model.py -q -m gpu 2>&1 | tail -5
