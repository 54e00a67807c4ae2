set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
echo "bench done"
for pol in bf16 fp8 fp4; do
  rm -rf gpurun_out/prof_r04f_$pol
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04f_$pol -- python3 bench.py --steps 16 --warmup 4 --policies $pol --no-cpu --no-gpt2 > gpurun_out/r04_bench_${pol}_under_rocprof.json 2> gpurun_out/prof_r04f_$pol.err
  python3 tools/summarize_rocprof.py gpurun_out/prof_r04f_$pol gpurun_out/r04_bench_${pol}_kernels.md "round 4: python bench.py --steps 16 --warmup 4 --policies $pol --no-cpu --no-gpt2 under rocprofv3 --kernel-trace --stats"
  rm -rf gpurun_out/prof_r04f_$pol
  echo "trace $pol done"
done
