#!/bin/bash
# usage: bash examples/train_celeba.sh <N landmarks: 10|30|50> [N GPUs]
# One process per GPU (RCCL over xGMI); dataset under the celeba_data_dir of configs/paths/default.yaml.
K=$1; G=${2:-1}
if [ "$G" -gt 1 ]; then
  python -m torch.distributed.run --nnodes=1 --nproc-per-node "$G" --master-addr 127.0.0.1 scripts/train.py \
    --configs configs/paths/default.yaml configs/experiments/celeba-"$K"pts.yaml --ngpus "$G"
else
  python scripts/train.py --configs configs/paths/default.yaml configs/experiments/celeba-"$K"pts.yaml
fi
