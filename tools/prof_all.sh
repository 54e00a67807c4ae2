# round profiles (GPU box): kernel traces of the full bench (prefill + decode) and PMC traffic passes, one policy at a time
set -e
tag=${1:-r1g}
policies=${2:-"bf16 fp8 fp4"}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for pol in $policies; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_$pol -- python3 bench.py --steps 16 --warmup 4 --policies $pol --no-cpu --no-gpt2 > gpurun_out/prof_${tag}_$pol.log 2>&1
  echo "trace $pol done"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcf_${tag}_$pol -- python3 bench.py --steps 4 --warmup 2 --policies $pol --no-cpu --no-gpt2 --no-prefill --mode fused > gpurun_out/pmcf_${tag}_$pol.log 2>&1
  echo "fetch $pol done"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcw_${tag}_$pol -- python3 bench.py --steps 4 --warmup 2 --policies $pol --no-cpu --no-gpt2 --no-prefill --mode fused > gpurun_out/pmcw_${tag}_$pol.log 2>&1
  echo "write $pol done"
done
