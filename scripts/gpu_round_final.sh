#!/bin/bash
# Everything the round's final state is judged on, in one gpurun call:  TAG=r04 bash scripts/gpu_round_final.sh
#   gpu_final.sh (driver's bench line, rocprofv3 kernel stats, calibrated PMC traffic), SQ counters of every kernel,
#   the forced-sharded step, the other configs' single-GPU legs.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${TAG:-r05}
export TMPDIR=/tmp
TAG=$TAG bash scripts/gpu_final.sh > gpurun_out/${TAG}_final.log 2>&1; tail -30 gpurun_out/${TAG}_final.log
TAG=${TAG}_sq bash scripts/gpu_pmc_kernels.sh > gpurun_out/${TAG}_sq.log 2>&1; tail -3 gpurun_out/${TAG}_sq.log
O=gpurun_out/${TAG}_final
MGC_BENCH_FORCE_SHARDED=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --no-db > $O/forced_sharded_bench.json 2> $O/forced_sharded_bench.err
echo "forced sharded exit $?"; python -c "import json; d=json.load(open('$O/forced_sharded_bench.json')); print('forced sharded ms/step', round(d['ms_per_step'],1))"
for a in "k21 21" "k31 31" "k31c 31 250000 1 20000" "k51 51" "k31c10 31 500000 1 20000" "k21rep10 21 66666667 0 150 100000" "k21sat10 21 66666667 0 150 30000 0 171 1"; do set -- $a; n=$1; shift
  timeout 300 python scripts/kbench.py "$@" > $O/kb_$n.json 2> $O/kb_$n.err
  python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['stage_ms_per_step'].items()})" $O/kb_$n.json
done
cat $O/kb_k21.json $O/kb_k31.json $O/kb_k31c.json $O/kb_k51.json $O/kb_k31c10.json $O/kb_k21rep10.json $O/kb_k21sat10.json > $O/kbench.jsonl
TAG=$TAG bash scripts/gpu_pmc_legs.sh > gpurun_out/${TAG}_pmc_legs.log 2>&1; tail -25 gpurun_out/${TAG}_pmc_legs.log
