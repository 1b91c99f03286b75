#!/bin/bash
# scratch: variants of the current build
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 900 python -u -m pytest tests/test_gpu_spgemm.py -x -q --timeout 300 --timeout-method thread -k "dense or split or default" > gpurun_out/exp_tests.log 2>&1 || { tail -15 gpurun_out/exp_tests.log; exit 1; }
tail -2 gpurun_out/exp_tests.log
RUNVAR_TIMEOUT=900 bash tools/runvar.sh gt24 24 SPMM_NOP=1 || exit 1
