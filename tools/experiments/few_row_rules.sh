# usage: bash tools/experiments/few_row_rules.sh <policy>  -> gpurun_out/few_row_rules.txt
# prefill ms of 2 .. 17-token prompts by the row count up to which the skinny (weight-streaming) kernels go ahead of the tile grids / split-K (gemm.skinny_ahead_rows = n, gemm.splitk_min_rows = n + 1)
set -e
pol=${1:-bf16}
out=gpurun_out/few_row_rules.txt
: > $out
for n in 16 8 4 1; do      # 1 = default
  echo "# skinny up to $n rows ($pol)" >> $out
  RAGGED_TUNE=gemm.skinny_ahead_rows=$n,gemm.splitk_min_rows=$((n + 1)) RAGGED_T=2,4,8,12,16,17,2049 timeout -k 10 300 python3 tools/bench_ragged_prefill.py $pol | tail -1 >> $out
done
cat $out
