#!/bin/bash
# contraction in the PLAIN skinning backward's transform blend only (-DDPOSER_CT2) vs the shipped library: LBS forward + backward at 4096 poses
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for rep in 1 2 3 4; do
  for v in shipped ct2; do
    unset DPOSER_LIB_PATH
    if [ $v = ct2 ]; then export DPOSER_LIB_PATH=$R/tools/bin/libdposer_hip_ct2.so; fi
    echo "## $v (run $rep): $(python3 tools/lbs_fwd_bwd_time.py 4096 2>&1 | grep 'LBS fwd')"
  done
done
