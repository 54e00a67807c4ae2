#!/bin/bash
# experiment driver (GPU box): prefill under the GEMM schedules
mkdir -p gpurun_out
for sch in ${SCHEDS:-0 1 3}; do
python bench.py --no-cpu --steps 8 --warmup 2 --gemm-schedule $sch "$@" > gpurun_out/sched_$sch.json 2>gpurun_out/sched_$sch.err
python - <<PY
import json
d=json.loads(open('gpurun_out/sched_$sch.json').read().strip().splitlines()[-1])
print($sch, {k:(v['prefill_ms'], v['prefill_TFLOPs']) for k,v in d['policies'].items()}, flush=True)
PY
done
