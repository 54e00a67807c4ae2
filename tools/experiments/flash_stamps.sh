#!/bin/bash
# Diagnostic build of the flash prefill kernel with in-kernel segment stamps (-DMILA_FLASH_STAMPS): a SEPARATE library, never the product one.
#   bash tools/experiments/flash_stamps.sh build     (here: cross-compiles)     then on the GPU box:   python tools/experiments/flash_stamps.py
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/experiments/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function ${FLASH_DEFS:--DMILA_FLASH_STAMPS} \
    -c mila_amd/csrc/attention_prefill.hip -o tools/experiments/_build/attention_prefill_stamps.o
objs=$(for f in mila_amd/csrc/*.hip; do b=$(basename $f .hip); [ $b != attention_prefill ] && echo mila_amd/lib/obj/$b.o; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/experiments/_build/libmila_cdna4_stamps.so $objs tools/experiments/_build/attention_prefill_stamps.o
echo built tools/experiments/_build/libmila_cdna4_stamps.so
