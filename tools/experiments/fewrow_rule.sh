# usage: bash tools/experiments/fewrow_rule.sh  -> gpurun_out/fewrow_rule.txt : bf16 / fp8-policy prefill ms at <= 33 rows (and a 1..32-row remainder) with the few-row
# weight-streaming form (csrc/gemm_fewrow_bf16.hip) off (gemm.fewrow = 0) and on (1)
set -e
out=gpurun_out/fewrow_rule.txt
: > $out
for v in 0 1; do
  echo "# gemm.fewrow=$v: few-row form $([ $v = 0 ] && echo off || echo on)" >> $out
  RAGGED_TUNE=gemm.fewrow=$v RAGGED_T=2,4,8,16,17,24,32,33,2049,2064,2080 timeout -k 10 400 python3 tools/bench_ragged_prefill.py bf16,fp8 | tail -1 >> $out
done
cat $out
