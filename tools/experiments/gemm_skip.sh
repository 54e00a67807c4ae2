#!/bin/bash
# Diagnostic build of the LDS-DMA GEMM with runtime switches that leave parts of the K loop out (-DMILA_GEMM_SKIP; env MILA_GEMM_SKIP = bit mask: 1 staging,
# 2 fragment reads, 4 MFMAs, 8 epilogue stores): a SEPARATE library, never the product one.  Results of such a run are garbage; only its time is read.
#   bash tools/experiments/gemm_skip.sh      (here: cross-compiles)     then on the GPU box:   MILA_GEMM_SKIP=<mask> python tools/experiments/gemm_skip.py
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/experiments/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -DMILA_GEMM_SKIP \
    -c mila_amd/csrc/gemm256.hip -o tools/experiments/_build/gemm256_skip.o
objs=$(for f in mila_amd/csrc/*.hip; do b=$(basename $f .hip); [ $b != gemm256 ] && echo mila_amd/lib/obj/$b.o; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/experiments/_build/libmila_cdna4_gemmskip.so $objs tools/experiments/_build/gemm256_skip.o
echo built tools/experiments/_build/libmila_cdna4_gemmskip.so
