#!/bin/bash
# Run on the GPU box:  bash tools/r06_small_nb_ab.sh -- the 128x32 tiling with two K-loop slots (tools/bin/libdposer_hip_nb2.so: the build before) vs three (shipped), interleaved
cd "$(dirname "$0")/.."
OLD=$PWD/tools/bin/libdposer_hip_nb2.so
for r in 1 2; do
  for b in 32 640 1280; do
    echo "train step B=$b NB2 (run $r): $(DPOSER_LIB_PATH=$OLD python tools/step_time.py --child $OLD 300 $b 2>/dev/null | grep MS)"
    echo "train step B=$b NB3 (run $r): $(python tools/step_time.py --child dposer_amd/libdposer_hip.so 300 $b 2>/dev/null | grep MS)"
  done
  echo "## NB2 (run $r)"; DPOSER_LIB_PATH=$OLD python tools/config_timings.py 2>/dev/null | grep "cfg3\|cfg5 \|cfg5 x 8 \|cfg4\|RK4, 25"
  echo "## NB3 (run $r)"; python tools/config_timings.py 2>/dev/null | grep "cfg3\|cfg5 \|cfg5 x 8 \|cfg4\|RK4, 25"
done
