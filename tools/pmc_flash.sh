# usage: bash tools/pmc_flash.sh <tag>   -> gpurun_out/pmc_flash_<tag>*/ (SQ counters of the flash prefill kernels on the two Gemma attention shapes, T = 2048), two counter passes
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_flash_${tag}_a -- python3 tools/bench_flash.py > gpurun_out/pmc_flash_${tag}_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_MISC --output-format csv -d gpurun_out/pmc_flash_${tag}_b -- python3 tools/bench_flash.py > gpurun_out/pmc_flash_${tag}_b.log 2>&1
python3 - <<PY
import csv, glob, collections
for part in "ab":
    f = glob.glob('gpurun_out/pmc_flash_${tag}_%s/**/*counter_collection.csv' % part, recursive=True)[0]
    g = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'flash' in r['Kernel_Name']:
            g[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, c in g.items():
        print('$tag', k, {n: round(sum(v) / len(v)) for n, v in c.items()})
PY
