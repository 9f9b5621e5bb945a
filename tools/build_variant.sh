#!/bin/bash
# Builds a variant of libumgen_hip.so with extra compile flags (kernel experiments): tools/build_variant.sh <name> <flags...>
#   -> umgen_amd/libumgen_hip_<name>.so ; select it at run time with UMGEN_LIB_PATH (never a fallback: same sources, other -D flags)
set -e
name=$1; shift
cd "$(dirname "$0")/../umgen_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form=1 \
  "-DUMGEN_SRC_HASH=\"variant-$name\"" "$@" -o ../libumgen_hip_$name.so \
  engine.hip gemm.hip gemm256.hip attn.hip gemv.hip oar_engine.hip decode_batched.hip rowops.hip frame.hip tokenizers.hip vqdec.hip debug_api.hip
echo built umgen_amd/libumgen_hip_$name.so
