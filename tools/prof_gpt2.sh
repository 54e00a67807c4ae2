# usage: bash tools/prof_gpt2.sh <tag>   -> gpurun_out/prof_<tag>_gpt2/ + gpurun_out/<tag>_gpt2_kernels.md
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_gpt2 -- python3 tools/bench_gpt2.py 3 > gpurun_out/prof_${tag}_gpt2.log 2>&1
python3 tools/summarize_rocprof.py gpurun_out/prof_${tag}_gpt2 gpurun_out/${tag}_gpt2_kernels.md "round ${tag}: rocprofv3 --kernel-trace --stats -- python3 tools/bench_gpt2.py 3 (GPT-2 124M bf16, B=8, T=1024)"
