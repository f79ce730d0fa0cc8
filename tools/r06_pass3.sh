#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_score.py -x -q -m gpu -k "timemlps or other_activations or other_model_configurations" 2>&1 | tail -25 > $O/r06_t7.log
cat $O/r06_t7.log
python tools/timemlps_time.py 2>&1 | tail -6 > $O/r06_timemlps_time.txt
cat $O/r06_timemlps_time.txt
python tools/fk_occupancy_sweep.py > $O/r06_fk_occupancy.md 2>&1
cat $O/r06_fk_occupancy.md
