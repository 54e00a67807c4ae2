# usage: bash tools/experiments/few_row_rules_fp4.sh  -> gpurun_out/few_row_rules_fp4.txt
# fp4-policy prefill ms by the row count from which the W4A8 GEMMs may split K (gemm_fp8.splitk_min_rows = n); below it the skinny fp8 kernel keeps the rows
set -e
out=gpurun_out/few_row_rules_fp4.txt
: > $out
for n in 65 33 17 2; do      # 33 = default
  echo "# fp8 split-K from $n rows on" >> $out
  RAGGED_TUNE=gemm_fp8.splitk_min_rows=$n RAGGED_T=2,8,16,32,64,65,100,300,511,2303 timeout -k 10 300 python3 tools/bench_ragged_prefill.py fp4 | tail -1 >> $out
done
cat $out
