#!/bin/bash
# build a VARIANT of the library for same-box A/B runs: tools/debug/build_variant.sh NAME "-DFOO=1 ..." -> tools/debug/libvar/NAME/libasm_hip.so
# (ASM_HIP_LIB=<that path> loads it; same ABI)
NAME=$1; FLAGS=$2
OUT=tools/debug/libvar/$NAME; mkdir -p $OUT/obj
for f in assembled_cnn_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc $FLAGS -c $f -o $OUT/obj/$b.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libasm_hip.so $OUT/obj/*.o && rm -rf $OUT/obj && ls -la $OUT/libasm_hip.so
