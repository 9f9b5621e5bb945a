#!/bin/bash
# Builds a variant of libumgen_hip.so with extra compile flags (kernel experiments): tools/build_variant.sh <name> <flags...>
#   -> umgen_amd/libumgen_hip_<name>.so ; select it at run time with UMGEN_LIB_PATH (never a fallback: same sources, other -D flags)
# VARIANT_SOURCES="a.hip b.hip" recompiles only those files with the flags and links the shipped build's cached objects of the others
# (default: every source).
set -e
name=$1; shift
cd "$(dirname "$0")/../umgen_amd/csrc"
ALL="engine.hip gemm.hip gemm256.hip attn.hip gemv.hip oar_engine.hip oar_engine_wide.hip decode_batched.hip rowops.hip frame.hip tokenizers.hip vqdec.hip debug_api.hip"
SRC=${VARIANT_SOURCES:-$ALL}
objs=""
for f in $ALL; do
  if [[ " $SRC " == *" $f "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form=1 "-DUMGEN_SRC_HASH=\"variant-$name\"" "$@" -c $f -o /tmp/variant_${name}_$f.o &
    objs="$objs /tmp/variant_${name}_$f.o"
  else
    o=$(ls -t .obj/$f.*.o 2>/dev/null | head -1)
    [ -n "$o" ] || { echo "no cached object for $f: run build() first"; exit 1; }
    objs="$objs $o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libumgen_hip_$name.so $objs
echo built umgen_amd/libumgen_hip_$name.so
