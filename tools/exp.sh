#!/bin/bash
# scratch: variants of the current build
cd ${GRAFT_REPO_ROOT:-/root/repo}
D="SPMM_LIB_PATH=build/diag/libspmm_hip.so SPMM_BENCH_NOCHECK=1"
bash tools/runvar.sh dg0 22 $D SPMM_WIN_DBG=0 || exit 1
bash tools/runvar.sh dg_noid 22 $D SPMM_WIN_DBG=$((7<<8)) || exit 1
bash tools/runvar.sh dg_noval 22 $D SPMM_WIN_DBG=$((8<<8)) || exit 1
bash tools/runvar.sh dg_nost 22 $D SPMM_WIN_DBG=$((9<<8)) || exit 1
