#!/bin/bash
# Diagnosis builds: compile ONE kernel file with -D<MACRO>=<mask> for every mask given and link it with the objects of the production
# build into imm_amd/libimm_abl_<tag><mask>.so (load with IMM_HIP_LIB=<path>).  Results of such a library are wrong by construction;
# only kernel times are read (tools/bench_conv.py).   usage: tools/ablate_build.sh <file.hip> <MACRO> <tag> <mask>...
set -eu
cd "$(dirname "$0")/.."
f=$1; macro=$2; tag=$3; shift 3
python -c "from imm_amd import build; build.build(verbose=False)"
base=$(basename $f)
objs=$(ls imm_amd/build/*.hip.o | grep -v "/$base.o")
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DIMM_SOURCE_DIGEST='"variant"' -D$macro=$m -c imm_amd/csrc/$base -o imm_amd/build/abl_$tag$m.o &
done
wait
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o imm_amd/libimm_abl_$tag$m.so $objs imm_amd/build/abl_$tag$m.o
  echo built imm_amd/libimm_abl_$tag$m.so
done
