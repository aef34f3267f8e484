#!/bin/bash
# build a variant of libsxhip.so with extra compiler flags into smart-crossover_amd/variants/ (select it with SXHIP_LIB=...)
#   tools/build_variant.sh TAG "-DRB_RPL_V=2 -DRB_MINW_V=4"
set -e
TAG=$1; EXTRA=$2
R=$(cd "$(dirname "$0")/.." && pwd)/smart-crossover_amd
B=$R/variants/build_$TAG; mkdir -p $B
ls $R/csrc/*.hip | xargs -P 8 -I{} sh -c 'f={}; o='$B'/$(basename ${f%.hip}).o; hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-result -I'$R'/../include -I'$R'/csrc '"$EXTRA"' -c $f -o $o'
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/variants/libsxhip_$TAG.so $B/*.o
ls -la $R/variants/libsxhip_$TAG.so
