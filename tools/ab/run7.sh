cd $GRAFT_REPO_ROOT
for v in ATTN QKV; do SAGNN_LIB=$GRAFT_REPO_ROOT/sa-gnn_amd/lib/libsagnn_skip_$v.so timeout -k 10 300 python tools/ab/combine_test.py ref skip_$v 2>&1 | grep -E "d64_t16|d64_t8|d128_t16|d32_t16"; done
