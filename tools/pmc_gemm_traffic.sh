# usage: bash tools/pmc_gemm_traffic.sh <tag>  -> gpurun_out/pmc_gemm_traffic_<tag>.txt : HBM-side fetch / write bytes of the prefill GEMMs (separate --pmc passes), beside the
# bytes the operands hold (M = 2048).  Units as tools/summarize_pmc.py (MI355X_MICROARCH.md, HBM section, gfx950): fetched bytes = 2 x FETCH_SIZE x 1024, written bytes = WRITE_SIZE x 1024.
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_gemm_traffic_$tag.txt
: > $out
for shape in "3840 8192" "3840 30720" "4096 3840" "15360 3840"; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_tmp
    rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/pmc_tmp -- python3 tools/gemm_only.py $shape > gpurun_out/pmc_tmp.log 2>&1
    python3 - "$shape" $ctr >> $out <<PY
import csv, glob, sys
f = glob.glob('gpurun_out/pmc_tmp/**/*counter_collection.csv', recursive=True)[0]
v = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'gemm' in r['Kernel_Name'] and r['Counter_Name'] == sys.argv[2]]
K, N = map(int, sys.argv[1].split())
alg = {'FETCH_SIZE': (2048 * K + N * K) * 2, 'WRITE_SIZE': 2048 * N * 2}[sys.argv[2]]
avg = sum(v) / len(v)
mb = (2 * avg * 1024 if sys.argv[2] == 'FETCH_SIZE' else avg * 1024) / 1e6
print('K=%d N=%d %s: %.1f MB per launch (counter %.0f KB, %d launches); the operands hold %.1f MB' % (K, N, sys.argv[2], mb, avg, len(v), alg / 1e6))
PY
  done
done
cat $out
