mkdir -p gpurun_out/r6h
L='64,128,64,64;32,64,64,64;32,128,32,32;32,128,64,32'
for v in full h2a1 h2a2 h2a3; do
  if [ $v = full ]; then unset IMM_HIP_LIB; else export IMM_HIP_LIB=$PWD/imm_amd/libimm_abl_$v.so; fi
  echo "== $v"; timeout 120 python tools/bench_conv.py --layers "$L" 2>&1 | tail -5
done > gpurun_out/r6h/halo2_ablation.txt 2>&1
cat gpurun_out/r6h/halo2_ablation.txt
