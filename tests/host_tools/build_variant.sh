#!/usr/bin/env bash
# A variant library for an A/B on the GPU box: the product sources with extra -D flags.
#   tests/host_tools/build_variant.sh NAME -DLLMK_TK_...=N ...   ->  llm.f90_amd/csrc/variants/libllmk_NAME.so
# (select it with LLMK_LIB=...; *.so is git-ignored, the measurement goes to profiles/)
set -e
cd "$(dirname "$0")/../../llm.f90_amd"
name=$1; shift
mkdir -p csrc/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -Wno-unused-value -Wno-unused-result "$@" -shared csrc/llmk.hip \
  -o csrc/variants/libllmk_$name.so -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib 2>&1 | grep -E "error" || true
ls -la csrc/variants/libllmk_$name.so
