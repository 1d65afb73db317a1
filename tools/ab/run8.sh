cd $GRAFT_REPO_ROOT
for v in SCALAR_DOT SCALAR_W SCALAR_O; do echo "== $v"; SAGNN_LIB=$GRAFT_REPO_ROOT/sa-gnn_amd/lib/libsagnn_$v.so timeout -k 10 300 python tools/ab/dbg_d32.py 2>&1 | grep -v amdgpu.ids | cut -c1-150; done
