#!/bin/bash
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "narrowed or oversized or sparse or medium_scale or session_matches or repeat or two_digit or sort" 2>&1 | tail -3
MGC_GROUP_DBG=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-e2e --no-cpu-baseline 2>gpurun_out/r02u.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('ms', d['ms_per_step'], 'stages', d['stage_ms_per_step'], 'check', d.get('check', {}).get('ok'))
print('passA', r['avg_launch_ms'], 'passB', r['second_pass']['avg_launch_ms'])"
grep groupdbg gpurun_out/r02u.err
} > gpurun_out/r02u.log 2>&1
tail -12 gpurun_out/r02u.log
