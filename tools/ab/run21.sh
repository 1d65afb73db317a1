cd $GRAFT_REPO_ROOT
for i in 1 2; do
SAGNN_LIB=$PWD/scratch/ab/lstm_old.so python tools/ab/lstm_time2.py old 2>&1 | grep lstm
python tools/ab/lstm_time2.py carry 2>&1 | grep lstm
done
