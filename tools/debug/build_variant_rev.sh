#!/bin/bash
# build the library of another COMMIT's csrc/ + include/ for same-box A/B runs against the working tree (same ABI assumed):
#   tools/debug/build_variant_rev.sh NAME REV   ->  tools/debug/libvar/NAME/libasm_hip.so   (load with ASM_HIP_LIB=...; r6_ab_lib.sh)
NAME=$1; REV=$2
OUT=tools/debug/libvar/$NAME; rm -rf $OUT; mkdir -p $OUT/obj $OUT/tree
git archive $REV assembled_cnn_amd/csrc include | tar -x -C $OUT/tree
for f in $OUT/tree/assembled_cnn_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -c $f -o $OUT/obj/$b.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libasm_hip.so $OUT/obj/*.o && rm -rf $OUT/obj $OUT/tree && ls -la $OUT/libasm_hip.so
