#!/bin/bash
# Run on the GPU box:  bash tools/r06_fwd_tile_ab.sh  -- 256x256 (shipped) vs 128x128 tiles for the forward pose-blend GEMM (DPOSER_LBS_FWD_BIG=0), interleaved
cd "$(dirname "$0")/.."
for r in 1 2 3; do
  echo "## 256x256 (run $r)"; python tools/lbs_fwd_bwd_time.py 4096 7680 2>/dev/null | grep LBS
  echo "## 128x128 (run $r)"; DPOSER_LBS_FWD_BIG=0 python tools/lbs_fwd_bwd_time.py 4096 7680 2>/dev/null | grep LBS
done
