#!/bin/bash
# Run on the GPU box:  bash tools/r06_small64_ab.sh -- GroupNorm layers on 128x32 / 4 waves (DPOSER_SMALL64=0) vs 128x64 / 8 waves (shipped between 1024 and 2048 samples), interleaved
cd "$(dirname "$0")/.."
for b in 1280 1536 2048; do
  for r in 1 2 3; do
    echo "train step B=$b 128x32 (run $r): $(DPOSER_SMALL64=0 python tools/step_time.py --child dposer_amd/libdposer_hip.so 300 $b 2>/dev/null | grep MS)"
    echo "train step B=$b 128x64 (run $r): $(python tools/step_time.py --child dposer_amd/libdposer_hip.so 300 $b 2>/dev/null | grep MS)"
  done
done
for w in "1024 768" "1024 1024"; do set -- $w
  for b in $2; do for r in 1 2; do
    echo "train step B=$b 128x32 (run $r): $(DPOSER_SMALL64=0 python tools/step_time.py --child dposer_amd/libdposer_hip.so 300 $b 2>/dev/null | grep MS)"
    echo "train step B=$b 128x64 forced (run $r): $(DPOSER_SMALL64_MIN=512 python tools/step_time.py --child dposer_amd/libdposer_hip.so 300 $b 2>/dev/null | grep MS)"
  done; done
done
