# usage: bash tools/pmc_clock.sh <tag>   -> gpurun_out/pmc_clock_<tag>.txt
# The clock the chip holds inside the prefill GEMMs (MI355X_MICROARCH.md, DVFS give-back): GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration, and how much of those cycles
# the matrix cores are busy (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)).  Long back-to-back runs on random data; dispatches under 0.3 ms read high.
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_clock_$tag.txt
: > $out
run() {   # <label> <K> <N>
  rm -rf gpurun_out/pmc_tmp
  MILA_GEMM_LAUNCHES=600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d gpurun_out/pmc_tmp -- python3 tools/gemm_only.py $2 $3 > gpurun_out/pmc_tmp.log 2>&1
  python3 - "$1" >> $out <<PY
import csv, glob, collections, sys
f = glob.glob('gpurun_out/pmc_tmp/**/*counter_collection.csv', recursive=True)[0]
rows = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if 'gemm' in r['Kernel_Name']:
        d = rows[int(r['Dispatch_Id'])]
        d[r['Counter_Name']] = float(r['Counter_Value'])
        d['us'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        d['name'] = r['Kernel_Name'][:50]
last = [rows[k] for k in sorted(rows)][-100:]      # the sustained part of the run
n = len(last)
us = sum(d['us'] for d in last) / n
ghz = sum(d['GRBM_GUI_ACTIVE'] for d in last) / n / 8 / us / 1e3
mfma = sum(d['SQ_VALU_MFMA_BUSY_CYCLES'] for d in last) / (4 * sum(d['SQ_BUSY_CU_CYCLES'] for d in last))
print(sys.argv[1], last[-1]['name'], {'dispatches': n, 'us': round(us, 1), 'effective_clock_GHz': round(ghz, 3), 'mfma_busy_frac_of_cu_cycles': round(mfma, 3)})
PY
}
run "bf16 qkv K3840 N8192" 3840 8192
run "bf16 gate_up(plain) K3840 N30720" 3840 30720
run "bf16 K15360 N8192" 15360 8192
run "bf16 down K15360 N3840" 15360 3840
cat $out
