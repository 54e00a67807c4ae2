# usage: bash tools/pmc_gemm_fp8.sh <tag>   -> gpurun_out/pmc_gemm_fp8_<tag>.txt : SQ counters of the fp8 x fp8 LDS-DMA GEMMs on the four Gemma prefill shapes (M = 2048) beside the bf16 ones
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_gemm_fp8_$tag.txt
: > $out
C1="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS"
C2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_MISC"
run() {   # <label> <program + args>
  label=$1; shift
  for part in 1 2; do
    if [ $part = 1 ]; then C=$C1; else C=$C2; fi
    rm -rf gpurun_out/pmc_tmp
    rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc_tmp -- python3 "$@" > gpurun_out/pmc_tmp.log 2>&1
    python3 - "$label" >> $out <<PY
import csv, glob, collections, sys
f = glob.glob('gpurun_out/pmc_tmp/**/*counter_collection.csv', recursive=True)[0]
g = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if 'gemm' in r['Kernel_Name']:
        g[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in g.items():
    print(sys.argv[1], k, {n: round(sum(v) / len(v)) for n, v in c.items()})
PY
  done
}
run "fp8 qkv K3840 N8192" tools/gemm_fp8_only.py 3840 8192
run "fp8 gate_up+geglu K3840 N30720" tools/gemm_fp8_only.py 3840 30720 geglu
run "fp8 down K15360 N3840" tools/gemm_fp8_only.py 15360 3840
run "bf16 qkv K3840 N8192" tools/gemm_only.py 3840 8192
run "bf16 down K15360 N3840" tools/gemm_only.py 15360 3840
cat $out
