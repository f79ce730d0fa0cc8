#!/bin/bash
# the shipped build (contraction where vertices are formed + in the TMP pre-pass) against the round-6 build before it (tools/bin/libdposer_hip_r06a.so)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
python -m pytest tests/test_gpu_fk.py tests/test_gpu_tasks.py tests/test_gpu_assets.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -4
OLD=$R/tools/bin/libdposer_hip_r06a.so
for rep in 1 2 3; do
  for v in before shipped; do
    echo "## $v (run $rep)"
    if [ $v = before ]; then export DPOSER_LIB_PATH=$OLD; else unset DPOSER_LIB_PATH; fi
    python3 tools/lbs_fwd_bwd_time.py 4096 7680 2>&1 | grep "LBS fwd"
    python3 tools/lbs_ab.py 2>&1 | grep "n=  4096" | tail -1
    python3 tools/config_timings.py cfg5 fused-only 2>&1 | grep "cfg5 x"
  done
done
