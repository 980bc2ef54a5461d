#!/bin/bash
# usage: bash examples/test_aflw.sh <N landmarks>
python scripts/test.py --experiment-name aflw-"$1"pts-finetune --train-dataset aflw --test-dataset aflw
