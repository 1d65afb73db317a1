cd $GRAFT_REPO_ROOT
for b in 512 384 256 192 128; do echo blocks $b; SAGNN_AB_MHSA_BLOCKS=$b timeout -k 10 300 python tools/ab/overlap_mhsa.py 2>&1 | tail -1; done
