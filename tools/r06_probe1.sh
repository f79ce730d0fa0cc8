#!/bin/bash
# round 6, first measurement pass (one GPU box): cfg 5 legs, L2 stream probe, RCCL one-rank latency, kernel trace of the batched motion-denoising loop
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_configs.py -q -m gpu -k cfg5 -s 2>&1 | grep -E "cfg5|passed|failed" > $O/r06_cfg5_legs.txt
$R/tools/bin/l2_stream_probe > $O/r06_l2_stream_probe.md 2>&1
python3 tools/rccl_latency.py > $O/r06_rccl_latency.md 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_md
rocprofv3 --kernel-trace --stats -d $O/prof_md -o md -- python3 $R/tools/md_prof.py > /dev/null 2>&1
DB=$(find $O/prof_md -name "*.db" | head -1)
python3 $R/tools/rocpd_summary.py $DB > $O/r06_md_stats_before.md 2>&1
rm -rf $O/prof_md
cat $O/r06_cfg5_legs.txt $O/r06_l2_stream_probe.md $O/r06_rccl_latency.md; head -40 $O/r06_md_stats_before.md
