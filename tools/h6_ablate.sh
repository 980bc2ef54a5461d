#!/bin/bash
# Diagnosis builds of conv_hdeep6.hip: for every bit mask in "$@" compile ONLY that file with -DIMM_H6_ABLATE=<mask> and link it with
# the objects of the production build into imm_amd/build/libimm_h6a<mask>.so (load it with IMM_HIP_LIB=<path>).  Results of such a
# library are wrong by construction; only kernel times are read (tools/bench_conv.py).  Run on the build container.
set -eu
cd "$(dirname "$0")/.."
python -c "from imm_amd import build; build.build(verbose=False)"
objs=$(ls imm_amd/build/*.hip.o | grep -v conv_hdeep6)
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DIMM_SOURCE_DIGEST='"variant"' -DIMM_H6_ABLATE=$m \
      -c imm_amd/csrc/conv_hdeep6.hip -o imm_amd/build/h6a$m.o &
done
wait
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o imm_amd/libimm_h6a$m.so $objs imm_amd/build/h6a$m.o
  echo built imm_amd/libimm_h6a$m.so
done
