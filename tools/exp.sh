#!/bin/bash
# scratch: variants of the current build
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 600 python tools/shard_balance.py 24 8 1 > gpurun_out/v_bal24_q.jsonl 2> gpurun_out/v_bal24_q.err || { tail -3 gpurun_out/v_bal24_q.err; exit 1; }
python3 -c "
import json
j=json.loads(open('gpurun_out/v_bal24_q.jsonl').read())
print('quarter free', j['balance'], [r['ms'] for r in j['ranks']])"
bash tools/runvar.sh q23 23 SPMM_NOP=1 || exit 1
RUNVAR_TIMEOUT=900 bash tools/runvar.sh q24 24 SPMM_NOP=1 || exit 1
