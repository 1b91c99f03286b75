#!/bin/bash
# usage: runvar.sh name scale ENV=... ENV=...
name=$1; scale=$2; shift 2
env "$@" timeout -k 10 ${RUNVAR_TIMEOUT:-500} python bench.py --scale $scale --steps 1 --warmup 1 --no-cpu > gpurun_out/v_$name.json 2> gpurun_out/v_$name.err
python - <<PY
import json
try:
    j=json.loads(open('gpurun_out/v_$name.json').read().strip().splitlines()[-1])
    p=j['phases_ms_per_step']
    print('$name', 'scale $scale', j['value'],'GF', j['ms_per_step'],'ms', 'dense_num',p['dense_num'],'dense_sym',p['dense_sym'],'win',p['window_num'],'light',p['light_num'],'wsym',p['window_sym'],'plan',p['slab_plan'],'bins',p['bins'], 'numfrac', j['phase_totals_rank0']['numeric']['frac_of_hbm_peak'])
except Exception as e:
    print('$name ERR', e, open('gpurun_out/v_$name.err').read()[-500:])
    raise SystemExit(1)
PY
