cd $GRAFT_REPO_ROOT
for i in 1 2; do
SAGNN_LIB=$PWD/scratch/ab/base.so python tools/ab/adam_time.py base 2>&1 | grep adam
python tools/ab/adam_time.py vec4 2>&1 | grep adam
done
timeout -k 10 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train.py -x -q -k "adam or Adam or train" 2>&1 | tail -3
