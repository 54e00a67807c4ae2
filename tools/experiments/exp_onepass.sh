#!/bin/bash
# experiment driver (GPU box): one-pass attention and prefetch-ahead variants of the decode bench
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_fused_gpu.py tests/test_gemma_host_gpu.py -x -q -k "onepass or prefetch" > gpurun_out/exp_tests.log 2>&1 || { tail -30 gpurun_out/exp_tests.log; exit 1; }
tail -3 gpurun_out/exp_tests.log
B="python bench.py --no-cpu --no-prefill --steps 96 --warmup 16"
$B > gpurun_out/exp_base.json 2>gpurun_out/exp_base.err && echo base done
$B --onepass 1 > gpurun_out/exp_onepass.json 2>gpurun_out/exp_onepass.err && echo onepass done
$B --onepass 1 --prefetch-mb 32 > gpurun_out/exp_pf32.json 2>gpurun_out/exp_pf32.err && echo pf32 done
$B --onepass 1 --prefetch-mb 96 > gpurun_out/exp_pf96.json 2>gpurun_out/exp_pf96.err && echo pf96 done
$B --onepass 1 --prefetch-mb 32 --prefetch-wgs 16 > gpurun_out/exp_pf32w16.json 2>gpurun_out/exp_pf32w16.err && echo pf32w16 done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/exp_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, {k:(v['tok_s'], v['dominant_kernel']['avg_us']) for k,v in d['policies'].items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
