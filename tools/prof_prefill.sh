# usage: bash tools/prof_prefill.sh <tag> <policy>...   -> gpurun_out/prof_<tag>_<policy>/ (bench with the T=2048 prefill, few decode steps)
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for pol in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_$pol -- python3 bench.py --steps 4 --warmup 2 --policies $pol --no-cpu > gpurun_out/prof_${tag}_$pol.log 2>&1
  echo "trace $pol done"
done
