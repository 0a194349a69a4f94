#!/bin/bash
# A differently COMPILED library for an A/B (ab_lib.sh / ab_ops_lib.sh): bash profiles/scripts/build_variant.sh <name> "<-D flags>"
# The 16-bit engines' translation units (bf16 and fp16 storage) are rebuilt with the flags; output achelous_amd/csrc/build/variants/lib_<name>.so (git-ignored, travels with gpurun).
set -e
cd "$(dirname "$0")/../../achelous_amd/csrc"
[ -n "$NOMAKE" ] || make -s -j8 >/dev/null
mkdir -p build/variants
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function -Wno-unused-variable -Wno-missing-braces $2 -c engine_bf16.cpp -o build/variants/engine_bf16_$1.o &
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function -Wno-unused-variable -Wno-missing-braces $2 -c engine_f16.cpp -o build/variants/engine_f16_$1.o &
wait
F32=build/engine_f32.o
if [ -n "$ALLTU" ]; then   # flags that change a kernel both engines instantiate (same symbol in both code objects: the runtime registers the first)
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function -Wno-unused-variable -Wno-missing-braces $2 -c engine_f32.cpp -o build/variants/engine_f32_$1.o
  F32=build/variants/engine_f32_$1.o
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/lib_$1.so build/engine.o $F32 build/api.o build/variants/engine_bf16_$1.o build/variants/engine_f16_$1.o
echo build/variants/lib_$1.so
