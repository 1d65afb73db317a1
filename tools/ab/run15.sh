cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
SAGNN_LIB=$PWD/scratch/ab_notail/libsagnn_base.so python tools/ab/lstm_time.py base 2>&1 | grep lstm
SAGNN_LIB=$PWD/scratch/ab_notail/libsagnn_notail.so python tools/ab/lstm_time.py notail 2>&1 | grep lstm
