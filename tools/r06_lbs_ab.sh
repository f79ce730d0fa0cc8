#!/bin/bash
# round 6: LBS forward + backward at 4096 poses and the batched motion-denoising step with / without the row-concatenated blend-gradient launch
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_fk.py tests/test_gpu_tasks.py -x -q -m gpu 2>&1 | tail -6
{
echo "# row-concatenated blend-gradient launch (DPOSER_LBS_BWD_ROWCAT), interleaved on one box"
for rep in 1 2 3; do
  for v in 1 0; do
    echo "## DPOSER_LBS_BWD_ROWCAT=$v (run $rep)"
    DPOSER_LBS_BWD_ROWCAT=$v python3 tools/lbs_fwd_bwd_time.py 2>&1 | grep "LBS fwd"
    DPOSER_LBS_BWD_ROWCAT=$v python3 tools/config_timings.py cfg5 fused-only 2>&1 | grep "cfg5 x"
  done
done
} > $O/r06_lbs_rowcat_ab.md
cat $O/r06_lbs_rowcat_ab.md
