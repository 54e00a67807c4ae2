# kernel-trace view of the attention pair for the two grids of tools/experiments/attn_pair.sh (baseline, attn.xcd_local=1): -> gpurun_out/attn_pair_kernels.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_attnpair_$v -- python3 bench.py --steps 16 --warmup 4 --policies bf16 --no-cpu --no-prefill --no-gpt2 --tune attn.xcd_local=$v > gpurun_out/prof_attnpair_$v.log 2>&1
  python3 tools/summarize_rocprof.py gpurun_out/prof_attnpair_$v gpurun_out/attnpair_$v.md "attn.xcd_local=$v" > /dev/null
  echo "## attn.xcd_local=$v" >> gpurun_out/attn_pair_kernels.txt
  grep -E "attn_decode_kernel|attn_combine" gpurun_out/attnpair_$v.md >> gpurun_out/attn_pair_kernels.txt
done
cat gpurun_out/attn_pair_kernels.txt
