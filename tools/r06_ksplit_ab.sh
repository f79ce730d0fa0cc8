#!/bin/bash
# blend-gradient GEMM split count (DPOSER_LBS_BWD_KSPLIT) A/B: LBS forward + backward at 4096 / 7680 poses, two interleaved rounds
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for rep in 1 2; do
  for f in 0 2 3 4 6 8 12; do
    echo "## DPOSER_LBS_BWD_KSPLIT=$f (run $rep)"
    DPOSER_LBS_BWD_KSPLIT=$f python3 tools/lbs_fwd_bwd_time.py 4096 7680 2>&1 | grep "LBS fwd"
  done
done
