#!/bin/bash
# experiment driver (GPU box): Linear / GEMM tests, then the bench with the T=2048 prefill for all policies
mkdir -p gpurun_out
python -m pytest tests/test_linear_gpu.py tests/test_gemma_host_gpu.py tests/test_gpt_host_gpu.py -x -q > gpurun_out/prefill_tests.log 2>&1 || { tail -30 gpurun_out/prefill_tests.log; exit 1; }
tail -2 gpurun_out/prefill_tests.log
python bench.py --no-cpu --steps 64 --warmup 8 "$@" > gpurun_out/prefill_bench.json 2>gpurun_out/prefill_bench.err
python - <<PY
import json
d=json.loads(open('gpurun_out/prefill_bench.json').read().strip().splitlines()[-1])
print({k:(v['tok_s'], v['prefill_ms'], v['prefill_TFLOPs']) for k,v in d['policies'].items()})
PY
