#!/bin/bash
# same-box A/B of two library builds on the convolution micro-benchmark: tools/ab_conv.sh <prev .so> [layer filter]
PREV=$1; F=${2:-vgg}
for i in 1 2; do
  echo "-- prev";  IMM_HIP_LIB=$PREV python tools/bench_conv.py 2>/dev/null | grep "$F"
  echo "-- new";   python tools/bench_conv.py 2>/dev/null | grep "$F"
done
