#!/bin/bash
# Builds the library for gfx950 in-tree, twice (the .so files are git-ignored but travel with gpurun):
#   libcontrolar_hip.so      the shipped library: no A/B or profiling switches (CAR_KNOB() is a null constant: neither getenv calls nor their names)
#   libcontrolar_hip_dev.so  the same sources with -DCAR_DEV_KNOBS: the CAR_* environment switches of DESIGN.md §4 (tools/*_sweep.py, tools/pmc_*.py, the
#                            schedule-invariance tests); loaded by controlar_amd._lib only when CONTROLAR_DEV_LIB=1 or Engine(..., dev=True)
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
SRCS="gemm ops decode decode2 decode_f32 pack canny t5 attn engine engine_weights engine_encode engine_generate engine_t5 engine_vq"
mkdir -p _obj _obj_dev
# build id = hash of every source: the packed-weight cache (car_export_packed / car_import_packed) is private to one build
BID=$( (cat *.hip *.h ../../include/controlar_hip.h; echo "$FLAGS"; $HIPCC --version 2>/dev/null) | sha1sum | cut -c1-40)
if [ ! -f _obj/build_id.h ] || ! grep -q "$BID" _obj/build_id.h; then echo "#define CAR_BUILD_ID \"$BID\"" > _obj/build_id.h; fi
cp -u _obj/build_id.h _obj_dev/build_id.h
pids=()
for variant in rel dev; do
  if [ $variant = rel ]; then O=_obj; X=""; else O=_obj_dev; X="-DCAR_DEV_KNOBS"; fi
  for f in $SRCS; do
    if [ ! -f $O/$f.o ] || [ $f.hip -nt $O/$f.o ] || [ car_common.h -nt $O/$f.o ] || [ decode2_params.h -nt $O/$f.o ] || [ kernel_params.h -nt $O/$f.o ] || [ decode_f32_params.h -nt $O/$f.o ] || { [ "${f#engine}" != "$f" ] && [ engine_internal.h -nt $O/$f.o ]; } || { [ $f = engine ] && [ _obj/build_id.h -nt $O/$f.o ]; } || [ ../../include/controlar_hip.h -nt $O/$f.o ]; then
      $HIPCC $FLAGS $X -I_obj -c $f.hip -o $O/$f.o &
      pids+=($!)
    fi
  done
done
for p in "${pids[@]}"; do wait $p; done
objs() { for f in $SRCS; do printf "%s/%s.o " $1 $f; done; }
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libcontrolar_hip.so $(objs _obj)
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libcontrolar_hip_dev.so $(objs _obj_dev)
echo "built $(pwd)/libcontrolar_hip.so and libcontrolar_hip_dev.so"
