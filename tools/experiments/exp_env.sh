#!/bin/bash
# experiment driver (GPU box): decode bench under runtime environment knobs of the HIP runtime
mkdir -p gpurun_out
B="python bench.py --no-cpu --no-prefill --steps 96 --warmup 16 --policies fp4"
run() { name=$1; shift; env "$@" $B > gpurun_out/env_$name.json 2>gpurun_out/env_$name.err; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/env_$name.json').read().strip().splitlines()[-1])
    print('$name', {k:v['tok_s'] for k,v in d['policies'].items()})
except Exception as e:
    print('$name', 'ERR', e)
PY
}
run base X=1
run optflush0 AMD_OPT_FLUSH=0
run optflush1 AMD_OPT_FLUSH=1
run devkernarg0 HIP_FORCE_DEV_KERNARG=0
run devkernarg1 HIP_FORCE_DEV_KERNARG=1
run pktcap0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run pktcap1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
# ROC_SYSTEM_SCOPE_SIGNAL=0 is NOT swept: it removes the system-scope completion signal the graph replay's host wait depends on and hangs the replay (round 1)
run kernargopt1 DEBUG_HIP_KERNARG_COPY_OPT=1
run hdpwa0 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0
run fgs ROC_USE_FGS_KERNARG=0
run direct0 AMD_DIRECT_DISPATCH=0
