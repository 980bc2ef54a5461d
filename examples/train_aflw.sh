#!/bin/bash
# usage: bash examples/train_aflw.sh <N landmarks> <celeba checkpoint: model.ckpt-N prefix (TensorFlow bundle) or .pt file>
# Fine-tunes the CelebA model on AFLW, optimizer state restored.
K=$1; CKPT=$2
python scripts/train.py --configs configs/paths/default.yaml configs/experiments/aflw-"$K"pts-finetune.yaml \
  --checkpoint "$CKPT" --restore-optim
