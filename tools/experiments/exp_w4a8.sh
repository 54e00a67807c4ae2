#!/bin/bash
# experiment driver (GPU box): W4A8 staging tests, then fp4 prefill under schedules 3 and 4
mkdir -p gpurun_out
python -m pytest tests/test_linear_gpu.py -x -q > gpurun_out/w4a8_tests.log 2>&1 || { tail -40 gpurun_out/w4a8_tests.log; exit 1; }
tail -2 gpurun_out/w4a8_tests.log
SCHEDS="3 4" bash tools/experiments/exp_sched.sh --policies fp4
