set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MILA_CDNA4_TUNING=1
rm -rf gpurun_out/prof_2k_mfma
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_2k_mfma -- python3 bench.py --steps 16 --warmup 4 --policies bf16 --no-cpu --no-gpt2 --no-prefill --attn-split -7 > gpurun_out/prof_2k_mfma.log 2>&1
python3 tools/summarize_rocprof.py gpurun_out/prof_2k_mfma gpurun_out/ctx2k_mfma_kernels.md "experiment: MFMA decode from a band of 1024 keys on"
