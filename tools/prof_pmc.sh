#!/bin/bash
# PMC passes over one short bench run each (run on the GPU box through gpurun).
# usage: tools/prof_pmc.sh <outdir-under-gpurun_out> <passes: sq1,sq2,sq3,tcc1,tcc2,tcc3,tcc4,tcp1,tcp2> <bench args...>
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1; shift
PASSES=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
declare -A C
C[sq1]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
C[sq2]="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU"
C[tcc1]="FETCH_SIZE TCC_REQ_sum"
C[tcc2]="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
C[tcc3]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
C[tcc4]="TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum"
C[sq3]="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
C[tcp1]="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE"
C[tcp2]="TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
for name in ${PASSES//,/ }; do
  timeout -s KILL 300 rocprofv3 --pmc ${C[$name]} --output-format csv -d $OUT/$name -- python3 $R/bench.py "$@" > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/$name.log; exit 1; }
  echo "pass $name ok"
done
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
grep -A 40 "^k_win<1024, 8192" $OUT/summary.txt | awk '/^k_/ && NR>1 {exit} {print}'
