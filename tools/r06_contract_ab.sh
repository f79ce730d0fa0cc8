#!/bin/bash
# FMA contraction in the skinning kernels (tools/bin/libdposer_hip_skin_contract.so = the library built with -DDPOSER_SKIN_CONTRACT) vs the shipped one
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
CT=$R/tools/bin/libdposer_hip_skin_contract.so
echo "# tests with the contracted library"
DPOSER_LIB_PATH=$CT python -m pytest tests/test_gpu_fk.py tests/test_gpu_tasks.py tests/test_gpu_assets.py -x -q -m gpu 2>&1 | tail -4
for rep in 1 2 3; do
  for v in shipped contract; do
    echo "## $v (run $rep)"
    if [ $v = contract ]; then export DPOSER_LIB_PATH=$CT; else unset DPOSER_LIB_PATH; fi
    python3 tools/lbs_fwd_bwd_time.py 4096 7680 2>&1 | grep "LBS fwd"
    python3 tools/config_timings.py cfg5 fused-only 2>&1 | grep "cfg5 x"
  done
done
