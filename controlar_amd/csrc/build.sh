#!/bin/bash
# Builds libcontrolar_hip.so for gfx950 in-tree (the .so is git-ignored but travels with gpurun).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
mkdir -p _obj
# build id = hash of every source: the packed-weight cache (car_export_packed / car_import_packed) is private to one build
BID=$( (cat *.hip *.h ../../include/controlar_hip.h; echo "$FLAGS"; $HIPCC --version 2>/dev/null) | sha1sum | cut -c1-40)
if [ ! -f _obj/build_id.h ] || ! grep -q "$BID" _obj/build_id.h; then echo "#define CAR_BUILD_ID \"$BID\"" > _obj/build_id.h; fi
pids=()
for f in gemm ops decode decode2 decode_f32 pack canny t5 attn engine engine_weights engine_encode engine_generate engine_t5 engine_vq; do
  if [ ! -f _obj/$f.o ] || [ $f.hip -nt _obj/$f.o ] || [ car_common.h -nt _obj/$f.o ] || [ decode2_params.h -nt _obj/$f.o ] || [ kernel_params.h -nt _obj/$f.o ] || [ decode_f32_params.h -nt _obj/$f.o ] || { [ "${f#engine}" != "$f" ] && [ engine_internal.h -nt _obj/$f.o ]; } || { [ $f = engine ] && [ _obj/build_id.h -nt _obj/$f.o ]; } || [ ../../include/controlar_hip.h -nt _obj/$f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o _obj/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libcontrolar_hip.so _obj/gemm.o _obj/ops.o _obj/decode.o _obj/decode2.o _obj/decode_f32.o _obj/pack.o _obj/canny.o _obj/t5.o _obj/attn.o _obj/engine.o _obj/engine_weights.o _obj/engine_encode.o _obj/engine_generate.o _obj/engine_t5.o _obj/engine_vq.o
echo "built $(pwd)/libcontrolar_hip.so"
