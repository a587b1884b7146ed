#!/bin/bash
# Experiment build: csrc/ + tail4.patch -> ab/libklnmf_tail4.so (the product sources are not touched).
#   bash experiments/tail4/build.sh      then on the GPU box:  KLNMF_LIB=ab/libklnmf_tail4.so KLNMF_ROW_TAIL4=1 python bench.py ...
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
T=$(mktemp -d /tmp/tail4_XXXX)
mkdir -p $T/multimodal_amd $R/ab
cp -r $R/multimodal_amd/csrc $T/multimodal_amd/ && cp -r $R/include $T/ && rm -f $T/multimodal_amd/csrc/*.so
(cd $T && patch -p0 < $R/experiments/tail4/tail4.patch)
cd $R && python3 - <<PY
import __graft_entry__ as g
g.CSRC = '$T/multimodal_amd/csrc'
g.compile_library('$R/ab/libklnmf_tail4.so')
print('built ab/libklnmf_tail4.so')
PY
rm -rf $T
