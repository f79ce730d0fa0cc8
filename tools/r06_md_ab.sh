#!/bin/bash
# round 6: the batched motion-denoising loop with the temporal term formed inside the skinning backward (default) against the round-5 form
# (DPOSER_MD_FUSED_TEMPORAL=1: k_skin_temporal + dverts through HBM), same box, interleaved; then a kernel trace of the new form
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
{
echo "# cfg 5 batched (tools/config_timings.py cfg5 fused-only), temporal term inside the skinning backward vs round-5 form, interleaved on one box"
for rep in 1 2; do
  for v in 2 1; do
    echo "## DPOSER_MD_FUSED_TEMPORAL=$v (run $rep)"
    DPOSER_MD_FUSED_TEMPORAL=$v python3 tools/config_timings.py cfg5 fused-only 2>&1 | grep "cfg5"
  done
done
} > $O/r06_md_ab.md
cat $O/r06_md_ab.md
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_md
rocprofv3 --kernel-trace --stats -d $O/prof_md -o md -- python3 $R/tools/md_prof.py > /dev/null 2>&1
DB=$(find $O/prof_md -name "*.db" | head -1)
python3 $R/tools/rocpd_summary.py $DB > $O/r06_md_stats_after.md 2>&1
rm -rf $O/prof_md
head -16 $O/r06_md_stats_after.md
