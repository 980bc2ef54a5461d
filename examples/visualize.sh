#!/bin/bash
# usage: bash examples/visualize.sh <experiment name> <directory with images> [output.png]
python scripts/visualize.py --experiment-name "$1" --images-dir "$2" --out "${3:-landmarks.png}"
