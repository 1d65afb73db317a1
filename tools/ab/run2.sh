set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SAGNN_LIB=$GRAFT_REPO_ROOT/sa-gnn_amd/lib/libsagnn_combine2.so timeout -k 10 300 python tools/ab/combine_test.py ref dpp > gpurun_out/r03_combine2.log 2>&1 && \
timeout -k 10 300 python tools/ab/combine_test.py check lds_vs_dpp >> gpurun_out/r03_combine2.log 2>&1
cat gpurun_out/r03_combine2.log
timeout -k 10 1000 python -m pytest tests/test_gpu_configs.py tests/test_gpu_multirank.py tests/test_gpu_backward.py tests/test_gpu_train.py tests/test_gpu_model.py tests/test_gpu_spmm.py -q -m gpu > gpurun_out/r03_pytest_b.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03_pytest_b.log
tail -40 gpurun_out/r03_pytest_b.log
