# usage: bash tools/prof_ctx.sh <context>  -> gpurun_out/ctx<context>_kernels.md (decode kernels at a long context, bf16, KV caches zero-filled)
set -e
ctx=${1:-32768}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_ctx$ctx
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ctx$ctx -- python3 bench.py --steps 16 --warmup 4 --policies bf16 --no-cpu --no-gpt2 --no-prefill --context $ctx > gpurun_out/prof_ctx$ctx.log 2>&1
python3 tools/summarize_rocprof.py gpurun_out/prof_ctx$ctx gpurun_out/ctx${ctx}_kernels.md "round 3: decode at context $ctx (bf16), rocprofv3 --kernel-trace --stats -- python3 bench.py --policies bf16 --no-prefill --context $ctx"
