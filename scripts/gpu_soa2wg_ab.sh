#!/bin/bash
# A/B of the two-workgroups-per-CU first pass (radix_group5_kernel, MGC_SOA_2WG): parity subset, then the timed step both ways.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${TAG:-r04aa}
O=gpurun_out/$TAG; mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "five_byte or hypothesis_grid" > $O/pytest.log 2>&1; echo "parity exit $?"; tail -3 $O/pytest.log | head -2
run() { n=$1; shift; env "$@" timeout 300 python bench.py --steps 5 --warmup 2 --no-e2e --no-cpu-baseline --no-db > $O/bench_$n.json 2> $O/bench_$n.err
  python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['stage_ms_per_step'].items()}, d.get('check',{}).get('ok'), 'pass1 ms', round(d['roofline']['sort_pass']['avg_launch_ms'],4))" $O/bench_$n.json; }
run wc    MGC_SOA_WC=1
run one   MGC_X=0
run wc_b  MGC_SOA_WC=1
