#!/bin/bash
# Run on the GPU box:  bash tools/r06_small_tile_max_sweep.sh -- with three K-loop slots on the 128x32 tiling, up to which batch does it beat 128x128?  (DPOSER_SMALL_TILE_MAX)
cd "$(dirname "$0")/.."
for b in 1536 2048 2560 3072 4096; do
  for r in 1 2; do
    echo "train step B=$b 128x128 (run $r): $(DPOSER_SMALL_TILE_MAX=1280 python tools/step_time.py --child dposer_amd/libdposer_hip.so 300 $b 2>/dev/null | grep MS)"
    echo "train step B=$b 128x32  (run $r): $(DPOSER_SMALL_TILE_MAX=4096 python tools/step_time.py --child dposer_amd/libdposer_hip.so 300 $b 2>/dev/null | grep MS)"
  done
done
