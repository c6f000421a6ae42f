#!/bin/bash
# round 2, call S: the whole GPU suite on the narrowed passes
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r02s_tests.log 2>&1
tail -25 gpurun_out/r02s_tests.log
