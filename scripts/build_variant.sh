#!/bin/bash
# Build an experiment variant of the library for scripts/ab.sh:  scripts/build_variant.sh TAG [-DFLAG ...]
# -> ab/libklnmf_TAG.so (git-ignored, travels to the GPU box).  -DKL_DEV_BUILD keeps only the KT = 7 and KT = 16 kernels.
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; shift
mkdir -p $R/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -shared -fPIC "$@" \
    -o $R/ab/libklnmf_$TAG.so $R/multimodal_amd/csrc/klnmf_api.hip 2>&1 | grep -v "warning\|^$" | head -20
ls -la $R/ab/libklnmf_$TAG.so
