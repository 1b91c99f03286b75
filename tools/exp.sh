#!/bin/bash
# scratch: variants of the current build
cd ${GRAFT_REPO_ROOT:-/root/repo}
D="SPMM_LIB_PATH=build/diag/libspmm_hip.so SPMM_BENCH_NOCHECK=1 SPMM_DENSE_SPARSE=0 SPMM_DENSE_KEEP_MIB=0"
bash tools/runvar.sh loc0 22 $D SPMM_WIN_DBG=0 || exit 1
bash tools/runvar.sh loc11 22 $D SPMM_WIN_DBG=$((11<<8)) || exit 1
bash tools/runvar.sh loc12 22 $D SPMM_WIN_DBG=$((12<<8)) || exit 1
