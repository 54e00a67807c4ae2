# usage: bash tools/pmc_gemm.sh <K> <N> <schedule> <tag>   -> gpurun_out/pmc_gemm_<tag>/ (SQ counters of the GEMM kernel)
set -e
K=$1; N=$2; S=$3; tag=$4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_gemm_$tag -- python3 tools/gemm_only.py $K $N $S > gpurun_out/pmc_gemm_$tag.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_gemm_$tag/**/*counter_collection.csv', recursive=True)[0]
g = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if 'gemm' in r['Kernel_Name']:
        g[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in g.items():
    print('$tag', k, {n: round(sum(v) / len(v)) for n, v in c.items()})
PY
