#!/bin/bash
# A/B of the pipelined count (MGC_PIPE): the timed step with and without it, then a parity subset.  TAG=r04y bash scripts/gpu_pipe_ab.sh
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${TAG:-r04y}
O=gpurun_out/$TAG; mkdir -p $O
run() { n=$1; shift; env "$@" timeout 300 python bench.py --steps 5 --warmup 2 --no-e2e --no-cpu-baseline --no-db > $O/bench_$n.json 2> $O/bench_$n.err
  python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['stage_ms_per_step'].items()}, d.get('check',{}).get('ok'))" $O/bench_$n.json; }
run pipe    MGC_X=0
run serial  MGC_PIPE=0
run pipe1s  MGC_FINISH_ALT=0
run ahead4  MGC_PIPE_AHEAD=4
run pipe_b  MGC_X=0
run serial_b MGC_PIPE=0
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "not full_size and not config4" > $O/pytest_parity.log 2>&1; echo "parity exit $?"; tail -3 $O/pytest_parity.log
