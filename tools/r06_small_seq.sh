#!/bin/bash
# Run on the GPU box:  bash tools/r06_small_seq.sh -- kernel sequences of one step of the small / latency-bound configurations:
#   cfg 3 (sampler step at 500 samples), cfg 5 (motion-denoising step of ONE 60-frame sequence), cfg 4 (completion step at 16384 poses)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/small_seq; rm -rf $O
rocprofv3 --kernel-trace --stats -d $O/c3 -o c3 -- python3 $R/tools/config_timings.py cfg3 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/c5 -o c5 -- python3 $R/tools/config_timings.py cfg5 one > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/c4 -o c4 -- python3 $R/tools/config_timings.py cfg4 > /dev/null 2>&1
cd $R
python3 tools/rocpd_summary.py --sequence $(find $O/c3 -name "*.db") EpiEmStep 700 > gpurun_out/r06_seq_cfg3.md
python3 tools/rocpd_summary.py --sequence $(find $O/c5 -name "*.db") k_md_update 300 > gpurun_out/r06_seq_cfg5.md
python3 tools/rocpd_summary.py --sequence $(find $O/c4 -name "*.db") ${CFG4_ANCHOR:-k_completion_update} 150 > gpurun_out/r06_seq_cfg4.md
python3 tools/rocpd_summary.py $(find $O/c4 -name "*.db") | head -30 > gpurun_out/r06_stats_cfg4.md
cat gpurun_out/r06_seq_cfg5.md gpurun_out/r06_seq_cfg4.md gpurun_out/r06_stats_cfg4.md
rm -rf $O
