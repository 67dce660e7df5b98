#!/bin/bash
# Build a variant of the library that differs only in npp_render.hip (extra hipcc flags after the output name), linking the
# other objects of the last full build:   tools/render_variant.sh build_ab/libnpp_rstamps.so -DNPP_RENDER_STAMPS
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$1; shift
mkdir -p "$(dirname "$OUT")"
OBJ=${OUT%.so}_render.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
    -Wno-unused-value -Wno-unused-result -I "$ROOT/include" "$@" -c "$ROOT/nclone_amd/csrc/npp_render.hip" -o "$OBJ"
OTHERS=$(ls "$ROOT"/nclone_amd/_obj/*.o | grep -v npp_render.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ" $OTHERS
echo "$OUT"
