#!/bin/bash
# PMC passes over one short bench run each (run on the GPU box through gpurun).
# usage: tools/prof_pmc.sh <outdir-under-gpurun_out> <bench args...>
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift; ctrs=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/$name -- python3 $R/bench.py "$@" > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/$name.log; return 1; }
  echo "pass $name ok"
}
pass sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "$@" &&
pass sq2 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" "$@" &&
pass tcc1 "FETCH_SIZE TCC_REQ_sum" "$@" &&
pass tcc2 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "$@"
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
