#!/bin/bash
# One launcher for the reference's example workflows (its examples/*.sh and visualize.ipynb):
#   examples/imm.sh train-celeba <K> [GPUS]        train K in {10,30,50} landmarks on CelebA (one process per GPU)
#   examples/imm.sh train-aflw   <K> <CKPT>        fine-tune on AFLW from a CelebA checkpoint (TF bundle prefix or .pt)
#   examples/imm.sh test-mafl    <K> [ITER]        landmark regression error on MAFL
#   examples/imm.sh test-aflw    <K> [ITER]        ... on AFLW
#   examples/imm.sh visualize    <EXPERIMENT> <IMAGE_DIR> [OUT.png]
set -e
cd "$(dirname "$0")/.."
PATHS=configs/paths/default.yaml
cmd=$1; shift || true
case "$cmd" in
  train-celeba)
    k=$1; g=${2:-1}; exp=configs/experiments/celeba-${k}pts.yaml
    if [ "$g" -gt 1 ]; then
      exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "$g" --master-addr 127.0.0.1 scripts/train.py --configs $PATHS "$exp" --ngpus "$g"
    fi
    exec python scripts/train.py --configs $PATHS "$exp" ;;
  train-aflw)
    exec python scripts/train.py --configs $PATHS configs/experiments/aflw-${1}pts-finetune.yaml --checkpoint "$2" --restore-optim ;;
  test-mafl)
    exec python scripts/test.py --experiment-name celeba-${1}pts --train-dataset mafl --test-dataset mafl ${2:+--iteration $2} ;;
  test-aflw)
    exec python scripts/test.py --experiment-name aflw-${1}pts-finetune --train-dataset aflw --test-dataset aflw ${2:+--iteration $2} ;;
  visualize)
    exec python scripts/visualize.py --experiment-name "$1" --images-dir "$2" --out "${3:-landmarks.png}" ;;
  *)
    sed -n 2,8p "$0"; exit 2 ;;
esac
