set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for pol in bf16 fp8 fp4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1f_$pol -- python3 bench.py --steps 16 --warmup 4 --policies $pol --no-cpu > gpurun_out/prof_f_$pol.log 2>&1
  echo "trace $pol done"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_$pol -- python3 bench.py --steps 4 --warmup 2 --policies $pol --no-cpu --no-prefill --mode fused > gpurun_out/pmc_f_$pol.log 2>&1
  echo "fetch $pol done"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_$pol -- python3 bench.py --steps 4 --warmup 2 --policies $pol --no-cpu --no-prefill --mode fused > gpurun_out/pmc_w_$pol.log 2>&1
  echo "write $pol done"
done
