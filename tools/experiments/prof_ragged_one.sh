# usage: bash tools/experiments/prof_ragged_one.sh <policy> <T>  -> gpurun_out/ragged_<policy>_<T>_kernels.md
set -e
pol=$1; T=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export RAGGED_T=$T
rm -rf gpurun_out/prof_ragged_one
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ragged_one -- python3 tools/bench_ragged_prefill.py $pol > gpurun_out/prof_ragged_one.log 2>&1
python3 tools/summarize_rocprof.py gpurun_out/prof_ragged_one gpurun_out/ragged_${pol}_${T}_kernels.md "$pol policy, prefill of $T tokens"
