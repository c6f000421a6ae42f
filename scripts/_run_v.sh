set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06v9; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "full_size_properties or config5_shape" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -2 $O/pytest.log
TAG=r06v9 AB="new:;v2048:MGC_PART_VGRID=2048" ROUNDS=4 bash scripts/gpu_ab.sh
