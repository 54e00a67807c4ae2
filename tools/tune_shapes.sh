# usage: bash tools/tune_shapes.sh <policy> "<MILA_MATVEC_SHAPE value>" ...   -> one bench line (tok/s) per value
pol=$1; shift
for v in "$@"; do
  MILA_MATVEC_SHAPE="$v" timeout -k 10 200 python bench.py --steps 64 --warmup 8 --policies $pol --no-cpu --no-prefill > gpurun_out/tune_tmp.log 2>&1
  echo "$pol [$v] $(grep -o '"tok_s": [0-9.]*' gpurun_out/tune_tmp.log)"
done
