#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
python -m pytest tests/test_gpu_fk.py tests/test_gpu_tasks.py tests/test_gpu_assets.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -12
OLD=$R/tools/bin/libdposer_hip_r06a.so
for rep in 1 2 3; do
  for v in before shipped shipped-g4; do
    echo "## $v (run $rep)"
    unset DPOSER_LIB_PATH DPOSER_SKIN_BWD_MFMA
    if [ $v = before ]; then export DPOSER_LIB_PATH=$OLD; fi
    if [ $v = shipped-g4 ]; then export DPOSER_SKIN_BWD_MFMA=4; fi
    python3 tools/lbs_fwd_bwd_time.py 4096 2>&1 | grep "LBS fwd"
    python3 tools/config_timings.py cfg5 fused-only 2>&1 | grep "cfg5 x"
  done
done
