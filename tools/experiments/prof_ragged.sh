# kernel durations of the fp4 policy's ragged prefills (tools/bench_ragged_prefill.py): which kernels serve the tails, and how fast
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_ragged
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ragged -- python3 tools/bench_ragged_prefill.py > gpurun_out/prof_ragged.log 2>&1
python3 tools/summarize_rocprof.py gpurun_out/prof_ragged gpurun_out/ragged_kernels.md "fp4 policy, ragged prefills (2048, 2049, 2000, 2303, 300, 16 tokens)"
