mkdir -p gpurun_out/r7b
L='64,128,64,64'
{ echo "== full"; timeout 100 python tools/bench_conv.py --layers "$L" 2>&1 | grep probe
for m in 1 2 3 8; do echo "== H2X_ABLATE=$m"; IMM_HIP_LIB=$PWD/imm_amd/libimm_x$m.so timeout 100 python tools/bench_conv.py --layers "$L" 2>&1 | grep probe; done
echo "== 16x16 form"; IMM_CONV_DISABLE=halo2x timeout 100 python tools/bench_conv.py --layers "$L" 2>&1 | grep probe; } | tee gpurun_out/r7b/ablate.txt
