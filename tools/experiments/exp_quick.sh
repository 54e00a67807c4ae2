#!/bin/bash
# experiment driver (GPU box): attention-related tests, then the decode bench for all policies
mkdir -p gpurun_out
python -m pytest tests/test_fused_gpu.py tests/test_attention_gpu.py tests/test_gemma_host_gpu.py -x -q > gpurun_out/quick_tests.log 2>&1 || { tail -30 gpurun_out/quick_tests.log; exit 1; }
tail -2 gpurun_out/quick_tests.log
python bench.py --no-cpu --no-prefill --steps 128 --warmup 16 "$@" > gpurun_out/quick_bench.json 2>gpurun_out/quick_bench.err
python - <<PY
import json
d=json.loads(open('gpurun_out/quick_bench.json').read().strip().splitlines()[-1])
print({k:(v['tok_s'], v['token_roofline_frac']) for k,v in d['policies'].items()})
PY
