#!/bin/bash
# scratch: variants of the current build
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/runvar.sh l128 22 SPMM_LIB_PATH=build/l128/libspmm_hip.so || exit 1
bash tools/runvar.sh l128_24 24 SPMM_LIB_PATH=build/l128/libspmm_hip.so || exit 1
