#!/bin/bash
# Run on the GPU box:  bash tools/small_step_pmc.sh <tag> [batch]
# SQ counters of every kernel of a small-batch training step (default 1280 poses, the reference's batch size): instructions issued,
# wave cycles, instruction-fetch and issue stalls -- what a latency-bound launch spends its microseconds on.
TAG=${1:-r04}
B=${2:-1280}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmc_${TAG}_small
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_IFETCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  -d $OUT/pmc_${TAG}_small -o s -- python3 $R/bench.py --global-batch $B --no-extra --no-cpu-baseline --no-live-roofline --steps 6 --warmup 3 > $OUT/${TAG}_small_pmc.log 2>&1
cd $R
{
  echo "# SQ counters per kernel, training step at $B poses ($TAG, commit $(cat $R/.head_rev 2>/dev/null || echo snapshot)); averages per launch"
  echo
  python3 tools/rocpd_summary.py --counters $(find $OUT/pmc_${TAG}_small -name "*.db")
  echo
  python3 tools/rocpd_summary.py $(find $OUT/pmc_${TAG}_small -name "*.db" | head -1) | head -40
} > $OUT/${TAG}_small_step_pmc_$B.md
rm -rf $OUT/pmc_${TAG}_small
cat $OUT/${TAG}_small_step_pmc_$B.md
