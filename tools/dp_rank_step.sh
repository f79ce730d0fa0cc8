#!/bin/bash
# Run on the GPU box:  bash tools/dp_rank_step.sh <tag>     (e.g. r04a)
# The per-rank data-parallel training step on ONE GPU: a one-rank "nccl" process group in which every collective really runs
# (DPOSER_DIST_FORCE_COLLECTIVES=1: bucket events -> communication stream -> ncclAllReduce), at the per-rank batches of the
# headline metric's 8 / 4 / 2-GPU legs (8192 / 16384 / 32768 poses), next to the plain single-process step of the same batch.
#   1. ms per step, forced collectives vs plain                       -> gpurun_out/<tag>_dp_rank_step.md
#   2. rocprofv3 kernel trace of the 8192 step (both forms) + 1280    -> gpurun_out/<tag>_seq_*.md
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
COMMON="--no-extra --no-cpu-baseline --no-live-roofline --steps 200 --warmup 20"
{
echo "# per-rank data-parallel step on one MI355X ($TAG, $(git -C $R rev-parse --short HEAD 2>/dev/null || cat $R/.head_rev 2>/dev/null || echo snapshot))"
echo
echo "| poses per rank | plain ms | forced one-rank RCCL group ms | extra env |"
echo "|---:|---:|---:|---|"
for B in 1280 4096 8192 16384 32768 65536; do
  P=$(python3 bench.py --global-batch $B $COMMON 2>/dev/null | grep '^{' | python3 -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
  F=$(WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 DPOSER_DIST_FORCE_COLLECTIVES=1 python3 bench.py --global-batch $B $COMMON 2>/dev/null | grep '^{' | python3 -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
  echo "| $B | $P | $F | ${DP_EXTRA_ENV:-} |"
done
} > $OUT/${TAG}_dp_rank_step.md
cat $OUT/${TAG}_dp_rank_step.md
cd /tmp && export TMPDIR=/tmp
SHORT="--no-extra --no-cpu-baseline --no-live-roofline --steps 12 --warmup 4"
for B in 8192 1280; do
  rm -rf $OUT/prof_${TAG}_p$B
  rocprofv3 --kernel-trace -d $OUT/prof_${TAG}_p$B -o t -- python3 $R/bench.py --global-batch $B $SHORT > /dev/null 2>&1
  DB=$(find $OUT/prof_${TAG}_p$B -name "*.db" | head -1)
  python3 $R/tools/rocpd_summary.py --sequence $DB k_prep_train 8 > $OUT/${TAG}_seq_plain_$B.md
  python3 $R/tools/rocpd_summary.py $DB > $OUT/${TAG}_stats_plain_$B.md
  rm -rf $OUT/prof_${TAG}_p$B
done
export WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 DPOSER_DIST_FORCE_COLLECTIVES=1
for B in 8192; do
  rm -rf $OUT/prof_${TAG}_f$B
  rocprofv3 --kernel-trace -d $OUT/prof_${TAG}_f$B -o t -- python3 $R/bench.py --global-batch $B $SHORT > /dev/null 2>&1
  DB=$(find $OUT/prof_${TAG}_f$B -name "*.db" | head -1)
  python3 $R/tools/rocpd_summary.py --sequence $DB k_prep_train 8 > $OUT/${TAG}_seq_forced_$B.md
  python3 $R/tools/rocpd_summary.py $DB > $OUT/${TAG}_stats_forced_$B.md
  rm -rf $OUT/prof_${TAG}_f$B
done
head -70 $OUT/${TAG}_seq_forced_8192.md
