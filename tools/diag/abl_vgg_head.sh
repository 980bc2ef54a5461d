mkdir -p gpurun_out/r6i
for v in full vha1 vha2 vha4; do
  if [ $v = full ]; then unset IMM_HIP_LIB; else export IMM_HIP_LIB=$PWD/imm_amd/libimm_abl_$v.so; fi
  echo "== $v"; timeout 120 python tools/diag/bench_vgg_head.py 2>&1 | grep bfloat16
done > gpurun_out/r6i/vgg_head_ablation.txt 2>&1
cat gpurun_out/r6i/vgg_head_ablation.txt
