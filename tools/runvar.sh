#!/bin/bash
# usage: runvar.sh name scale ENV=... ENV=...
# One bench.py run of a variant (environment switches of the library, another build through SPMM_LIB_PATH); the
# settings are written into the .json/.err it leaves under gpurun_out/ so that a result can be traced to its variant.
name=$1; scale=$2; shift 2
echo "runvar $name scale $scale env: $*" > gpurun_out/v_$name.err
env "$@" timeout -k 10 ${RUNVAR_TIMEOUT:-500} python bench.py --scale $scale --steps 1 --warmup 1 --no-cpu > gpurun_out/v_$name.json 2>> gpurun_out/v_$name.err
rc=$?
python - "$name" "$scale" "$rc" "$*" <<'PY'
import json, sys
name, scale, rc, env = sys.argv[1:5]
f = f'gpurun_out/v_{name}.json'
try:
    j = json.loads(open(f).read().strip().splitlines()[-1])
    j['variant'] = dict(name=name, scale=int(scale), env=env)
    open(f, 'w').write(json.dumps(j) + '\n')
    p = j['phases_ms_per_step']
    print(name, 'scale', scale, j['value'], 'GF', j['ms_per_step'], 'ms', 'dense_num', p['dense_num'], 'win', p['window_num'], 'light', p['light_num'],
          'wsym', p['window_sym'], 'esc', p['esc_sym'] + p['esc_num'], 'bins', p['bins'], 'numfrac', j['phase_totals_rank0']['numeric']['frac_of_hbm_peak'], '|', env)
except Exception as e:
    print(name, 'ERR rc', rc, e, open(f'gpurun_out/v_{name}.err').read()[-600:], '|', env)
    raise SystemExit(1)
PY
