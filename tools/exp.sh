#!/bin/bash
# scratch: variants of the current build
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/runvar.sh ts0 22 SPMM_TWO_STREAMS=0 || exit 1
bash tools/runvar.sh ts1 22 SPMM_TWO_STREAMS=1 || exit 1
RUNVAR_TIMEOUT=900 bash tools/runvar.sh ts1_24 24 SPMM_TWO_STREAMS=1 || exit 1
