set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06v8; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "gigantic or hypothesis or other_baseline or capacity" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -4 $O/pytest.log
TAG=r06v8 LEGS="k51sat5:51 33333334 0 150 30000 0 171 1;sat:21 66666667 0 150 30000 0 171 1;k31sat5:31 33333334 0 150 30000 0 171 1;k21rep10:21 66666667 0 150 100000;k51rep5:51 33333334 0 150 100000" AB="new:" ROUNDS=1 bash scripts/gpu_kab.sh
