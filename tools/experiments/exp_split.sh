#!/bin/bash
# experiment driver (GPU box): flash-decode split length x combine placement
mkdir -p gpurun_out
B="python bench.py --no-cpu --no-prefill --steps 96 --warmup 16"
run() { name=$1; shift; $B "$@" > gpurun_out/split_$name.json 2>gpurun_out/split_$name.err; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/split_$name.json').read().strip().splitlines()[-1])
    print('$name', {k:v['tok_s'] for k,v in d['policies'].items()}, flush=True)
except Exception as e:
    print('$name', 'ERR', e, flush=True)
PY
}
run base
run s128 --attn-split 128
run s256 --attn-split 256
run s128_fold --attn-split 128 --combine-in-oproj 1
run s256_fold --attn-split 256 --combine-in-oproj 1
run s64_fold --combine-in-oproj 1
