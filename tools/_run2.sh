set -u
mkdir -p gpurun_out/r5a
L='64,64,128,128;64,32,256,256;64,16,512,512'
for m in 0 1 2 4 3 7 8 16 23; do
  lib=imm_amd/libimm_h6a$m.so; [ $m = 0 ] && lib=imm_amd/libimm_hip.so
  echo "== ablate $m"; IMM_HIP_LIB=$PWD/$lib timeout 120 python tools/bench_conv.py --layers "$L" 2>&1 | grep probe
done | tee gpurun_out/r5a/h6_ablate.txt
