#!/bin/bash
# Build an experiment variant of the library for scripts/ab.sh:  scripts/build_variant.sh TAG [-DFLAG ...]
# -> ab/libklnmf_TAG.so (git-ignored, travels to the GPU box).  -DKL_DEV_BUILD keeps only the KT = 7 and KT = 16 kernels.
# Without flags: the product build (parallel translation units, __graft_entry__.compile_library); with flags: one unit.
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; shift
mkdir -p $R/ab
python3 $R/scripts/build_lib.py $R/ab/libklnmf_$TAG.so "$@" 2>&1 | grep -v "warning\|^$" | head -20
ls -la $R/ab/libklnmf_$TAG.so
