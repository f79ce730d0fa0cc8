#!/bin/bash
# Run on the GPU box:  bash tools/r06_silu_split_small_ab.sh -- the time-branch dgrad of the training step at <= 1280 poses (128 x 32 tiling): one launch walking
# K = 5 x 1024 per tile (DPOSER_SILU_SPLIT_MAX=0) vs one k-split per layer + the reduce pass (default since round 6 on this tiling too), interleaved
cd "$(dirname "$0")/.."
for b in 32 256 640 1280 2048; do
  for r in 1 2 3; do
    echo "B=$b one launch (run $r): $(DPOSER_SILU_SPLIT_MAX=0 python tools/step_time.py --child dposer_amd/libdposer_hip.so 300 $b 2>/dev/null | grep MS)"
    echo "B=$b k-split    (run $r): $(python tools/step_time.py --child dposer_amd/libdposer_hip.so 300 $b 2>/dev/null | grep MS)"
  done
done
