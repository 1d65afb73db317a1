set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_f16_range.py tests/test_gpu_dense.py tests/test_gpu_fusion_multitile.py tests/test_gpu_fusion.py -q -m gpu > gpurun_out/r03_pytest_a.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03_pytest_a.log
tail -30 gpurun_out/r03_pytest_a.log
timeout -k 10 300 python tools/ab/combine_test.py ref lds > gpurun_out/r03_combine.log 2>&1 && \
SAGNN_LIB=$GRAFT_REPO_ROOT/sa-gnn_amd/lib/libsagnn_combine1.so timeout -k 10 300 python tools/ab/combine_test.py check shfl >> gpurun_out/r03_combine.log 2>&1 && \
SAGNN_LIB=$GRAFT_REPO_ROOT/sa-gnn_amd/lib/libsagnn_combine2.so timeout -k 10 300 python tools/ab/combine_test.py check dpp >> gpurun_out/r03_combine.log 2>&1
cat gpurun_out/r03_combine.log
