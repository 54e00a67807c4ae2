#!/bin/bash
# experiment driver (GPU box): warm-ahead blocks inside the attention / combine launches
mkdir -p gpurun_out
python -m pytest tests/test_gemma_host_gpu.py tests/test_fused_gpu.py -x -q -k "onepass or prefetch" > gpurun_out/warm_tests.log 2>&1 || { tail -30 gpurun_out/warm_tests.log; exit 1; }
tail -2 gpurun_out/warm_tests.log
B="python bench.py --no-cpu --no-prefill --steps 96 --warmup 16"
run() { name=$1; shift; $B "$@" > gpurun_out/warm_$name.json 2>gpurun_out/warm_$name.err; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/warm_$name.json').read().strip().splitlines()[-1])
    print('$name', {k:v['tok_s'] for k,v in d['policies'].items()}, flush=True)
except Exception as e:
    print('$name', 'ERR', e, flush=True)
PY
}
run base
run a16 --warm 16,64,0,0
run a8 --warm 8,64,0,0
run a16b8 --warm 16,64,8,16
run a16b16 --warm 16,64,16,32
run a16c32b16 --warm 16,32,16,32
run a32b32 --warm 32,64,32,48
