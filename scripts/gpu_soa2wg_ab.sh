#!/bin/bash
# A/B of the first-pass kernels for 5-byte files: parity subset, then the timed step per variant.
#   TAG=r04ab VARIANTS="wc one wc one" bash scripts/gpu_soa2wg_ab.sh
# variants: one = default (radix_group_kernel<..., SOA>), two = MGC_SOA_2WG=1 (radix_group5_kernel), wc = MGC_SOA_WC=1 and
# wc2 = MGC_SOA_WC=2 (radix_group5wc_kernel<16, 32, 4> / <8, 16, 8>), pipe = MGC_PIPE=1 (pipelined count).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${TAG:-r04ab}
VARIANTS=${VARIANTS:-"wc one wc one"}
O=gpurun_out/$TAG; mkdir -p $O
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  timeout 500 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "five_byte or hypothesis_grid or pipelined" > $O/pytest.log 2>&1; echo "parity exit $?"; tail -3 $O/pytest.log | head -2
fi
i=0
for v in $VARIANTS; do i=$((i+1))
  case $v in one) e="MGC_X=0";; two) e="MGC_SOA_2WG=1";; wc) e="MGC_SOA_WC=1";; wc2) e="MGC_SOA_WC=2";; pipe) e="MGC_PIPE=1";; *) echo "unknown variant $v"; continue;; esac
  env $e timeout 300 python bench.py --steps 5 --warmup 2 --no-e2e --no-cpu-baseline --no-db > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err
  python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['stage_ms_per_step'].items()}, d.get('check',{}).get('ok'), 'pass1 ms', round(d['roofline']['sort_pass']['avg_launch_ms'],4))" $O/bench_${v}_$i.json
done
