#!/bin/bash
# usage: bash examples/test_mafl.sh <N landmarks>     (regressor trained on MAFL train, error on MAFL test)
python scripts/test.py --experiment-name celeba-"$1"pts --train-dataset mafl --test-dataset mafl
