#!/bin/bash
# The measurements behind DESIGN.md's figures, from the build in the tree (run on the GPU box through gpurun, in parts:
# tools/evidence.sh <tag> <part>; results land in gpurun_out/<tag>_*, the ones to keep are copied to profiles/).
# parts: bench | stats | pmc | class | balance | small | pre
set -u
TAG=$1; PART=$2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
cd $R
case $PART in
bench)
  timeout -k 10 600 python bench.py > $O/${TAG}_bench_rmat22.json 2> $O/${TAG}_bench22.err || { tail -3 $O/${TAG}_bench22.err; exit 1; }
  for s in 20 23 24; do
    n=$s; [ $s = 24 ] && n=24_one_gpu
    timeout -k 10 900 python bench.py --scale $s --steps 1 --warmup 1 --no-cpu > $O/${TAG}_bench_rmat$n.json 2> $O/${TAG}_bench$s.err || { tail -3 $O/${TAG}_bench$s.err; exit 1; }
  done
  python - <<PY
import json
for f in ("rmat20", "rmat22", "rmat23", "rmat24_one_gpu"):
    j = json.loads(open("$O/${TAG}_bench_%s.json" % f).read().strip().splitlines()[-1])
    print(f, j["value"], "GF", j["ms_per_step"], "ms", j["phases_ms_per_step"], "numeric", j["phase_totals_rank0"]["numeric"]["frac_of_hbm_peak"])
PY
  ;;
stats)
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/${TAG}_kt22
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt22 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu > $O/${TAG}_kt22.log 2>&1 || { tail -3 $O/${TAG}_kt22.log; exit 1; }
  db=$(find $O/${TAG}_kt22 -name "*.db" | head -1)
  python3 $R/tools/kernel_stats_from_db.py $db > $O/${TAG}_kernel_stats_rmat22.csv && head -12 $O/${TAG}_kernel_stats_rmat22.csv
  ;;
pmc)
  bash $R/tools/prof_pmc.sh ${TAG}_pmc22 sq1,sq2,sq3,tcc1,tcc2,tcc3,tcc4 --steps 1 --warmup 0 --no-cpu || exit 1
  cp $O/${TAG}_pmc22/summary.txt $O/${TAG}_pmc_rmat22_summary.txt
  CMD="tools/prof_pmc.sh ${TAG}_pmc22 sq1,sq2,sq3,tcc1,tcc2,tcc3,tcc4 --steps 1 --warmup 0 --no-cpu"
  python3 $R/tools/pmc_traffic.py $O/${TAG}_pmc22 "R-MAT scale 22 edge factor 16 seed 7, C=A*A, all values 1.0" "$CMD" > $O/${TAG}_pmc_traffic_rmat22.json
  python3 $R/tools/pmc_occupancy.py $O/${TAG}_pmc22 "$CMD" > $O/${TAG}_occupancy_rmat22.json
  python3 -c "import json; j=json.load(open('$O/${TAG}_pmc_traffic_rmat22.json')); print(j['source_digest'], {k: v['hbm_bytes_per_launch'] for k, v in j['kernels'].items()})"
  ;;
class)
  timeout -k 10 600 python tools/class_cost.py 22 > $O/${TAG}_row_class_cost_rmat22.jsonl 2> $O/${TAG}_class.err || { tail -3 $O/${TAG}_class.err; exit 1; }
  cat $O/${TAG}_row_class_cost_rmat22.jsonl
  ;;
balance)
  for a in "20 8" "22 8" "24 8"; do set -- $a
    timeout -k 10 900 python tools/shard_balance.py $1 $2 1 > $O/${TAG}_shard_balance_rmat$1.jsonl 2> $O/${TAG}_bal.err || { tail -3 $O/${TAG}_bal.err; exit 1; }
    python3 -c "import json; j=json.loads(open('$O/${TAG}_shard_balance_rmat$1.jsonl').read()); print('scale', j['scale'], 'seed', j['seed'], 'balance', j['balance'], 'projected', j['projected_gflops'])"
  done
  : > $O/${TAG}_shard_balance_held_out.jsonl
  for a in "22 8 1 11" "24 8 1 11" "askubuntu 8 2 0" "stanford 8 2 0"; do set -- $a
    timeout -k 10 900 python tools/shard_balance.py $1 $2 $3 $4 >> $O/${TAG}_shard_balance_held_out.jsonl 2> $O/${TAG}_bal.err || { tail -3 $O/${TAG}_bal.err; exit 1; }
  done
  python3 -c "
import json
for l in open('$O/${TAG}_shard_balance_held_out.jsonl'):
    j = json.loads(l); print('held out', j['scale'], 'seed', j['seed'], 'balance', j['balance'], 'projected', j['projected_gflops'])"
  ;;
small)
  timeout -k 10 600 python tools/bench_snap_like.py 5 > $O/${TAG}_snap_like_graphs.jsonl 2> $O/${TAG}_snap.err || { tail -3 $O/${TAG}_snap.err; exit 1; }
  cat $O/${TAG}_snap_like_graphs.jsonl | cut -c 1-400
  timeout -k 10 300 python tools/bench_spv8.py > $O/${TAG}_spv8_spmv_spmm.jsonl 2> $O/${TAG}_spv8.err || { tail -3 $O/${TAG}_spv8.err; exit 1; }
  cat $O/${TAG}_spv8_spmv_spmm.jsonl | cut -c 1-300
  ;;
pre)
  timeout -k 10 600 python tools/bench_preprocess.py 20 22 > $O/${TAG}_preprocess_gpu_vs_cpu.jsonl 2> $O/${TAG}_pre.err || { tail -3 $O/${TAG}_pre.err; exit 1; }
  cat $O/${TAG}_preprocess_gpu_vs_cpu.jsonl
  timeout -k 10 600 python tools/bench_mtx.py > $O/${TAG}_mtx_reader_gpu_vs_cpu.jsonl 2> $O/${TAG}_mtx.err || { tail -3 $O/${TAG}_mtx.err; exit 1; }
  cat $O/${TAG}_mtx_reader_gpu_vs_cpu.jsonl | cut -c 1-400
  ;;
esac
