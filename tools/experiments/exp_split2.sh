#!/bin/bash
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
run s32 --attn-split 32
run s48 --attn-split 48
run s96 --attn-split 96
