#!/bin/bash
# Run on the GPU box:  bash tools/r06_md_small_ab.sh -- motion denoising of ONE 60-frame sequence (cfg 5 as the reference runs it) and of 8:
#   before    tools/bin/libdposer_hip_r5.so (k_sum_slabs one load per round trip, three blend-gradient launches)
#   3 terms   shipped library, DPOSER_LBS_BWD_ROWCAT=2 (batched slab loads; three launches on the 128x128 tiles)
#   shipped   batched slab loads + the row-concatenated form on the 128x128 tiles too (two launches)
cd "$(dirname "$0")/.."
for r in 1 2 3; do
  echo "before  (run $r): $(DPOSER_LIB_PATH=$PWD/tools/bin/libdposer_hip_r5.so python tools/config_timings.py cfg5 fused-only 2>/dev/null | grep 'cfg5 |\|cfg5 x 8 ' | cut -d'|' -f2,5 | tr '\n' ' ')"
  echo "3 terms (run $r): $(DPOSER_LBS_BWD_ROWCAT=2 python tools/config_timings.py cfg5 fused-only 2>/dev/null | grep 'cfg5 |\|cfg5 x 8 ' | cut -d'|' -f2,5 | tr '\n' ' ')"
  echo "shipped (run $r): $(python tools/config_timings.py cfg5 fused-only 2>/dev/null | grep 'cfg5 |\|cfg5 x 8 ' | cut -d'|' -f2,5 | tr '\n' ' ')"
done
