cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
export LD_LIBRARY_PATH=$R/graphmat_amd
build/apps/swept_edge_values blocked 2>&1 | grep -v "GraphMat(HIP)" | tail -30
