#!/bin/bash
# experiment driver (GPU box): resident prefill staging test, then prefill with and without it
mkdir -p gpurun_out
python -m pytest tests/test_gemma_host_gpu.py tests/test_linear_gpu.py tests/test_capi_cpu.py -x -q > gpurun_out/resident_tests.log 2>&1 || { tail -40 gpurun_out/resident_tests.log; exit 1; }
tail -2 gpurun_out/resident_tests.log
for r in 0 1; do
python bench.py --no-cpu --steps 16 --warmup 4 --policies fp8,fp4 --resident $r > gpurun_out/resident_$r.json 2>gpurun_out/resident_$r.err
python - <<PY
import json
d=json.loads(open('gpurun_out/resident_$r.json').read().strip().splitlines()[-1])
print($r, {k:(v['prefill_ms'], v['prefill_TFLOPs'], v['tok_s']) for k,v in d['policies'].items()}, flush=True)
PY
done
