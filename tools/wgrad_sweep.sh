#!/bin/bash
# wgrad variants on the renderer / encoder layers: 32- vs 64-channel halo slices, map-size threshold
for cfg in "IMM_WGRAD_HALO_SLICE=32 IMM_WGRAD_HALO_SLICE_MIN=64" "IMM_WGRAD_HALO_SLICE=64 IMM_WGRAD_HALO_SLICE_MIN=64" "IMM_WGRAD_HALO_SLICE=6432 IMM_WGRAD_HALO_SLICE_MIN=64" "IMM_WGRAD_HALO_SLICE=64 IMM_WGRAD_HALO_SLICE_MIN=32" "IMM_WGRAD_HALO_SLICE=64 IMM_WGRAD_HALO_SLICE_MIN=16"; do
  echo "== $cfg"
  env $cfg python tools/bench_conv.py --wgrad 2>/dev/null | grep "wgrad" | grep "ren_conv\|enc_conv6\|enc_conv4"
done
