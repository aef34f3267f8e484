#!/bin/bash
# Builds an experimental variant of the library: tools/build_variant.sh NAME [extra hipcc flags]
# -> smart-crossover_amd/lib/libsxhip_NAME.so (select it with SXHIP_LIB=...); objects under build_NAME/.
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)/smart-crossover_amd
mkdir -p $R/build_$NAME $R/lib
for f in $R/csrc/*.hip; do
  o=$R/build_$NAME/$(basename ${f%.hip}).o
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-result \
        -I$R/../include -I$R/csrc "$@" -c $f -o $o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/lib/libsxhip_$NAME.so $R/build_$NAME/*.o
echo built $R/lib/libsxhip_$NAME.so
