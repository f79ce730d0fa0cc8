#!/bin/bash
# Run on the GPU box:  bash tools/fk_pmc.sh [tag]   -- rocprofv3 counter passes of the joints-only FK kernel at 2^20 poses: what, beside HBM,
# the kernel spends its time on (VALU issue, waits, LDS) and the bytes it moves
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
[ -f "$R/bench.py" ] || { echo "$0: $R is not the repository root" >&2; exit 2; }
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/fkpmc; rm -rf $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $O/m2 -o m -- python3 $R/tools/fk_prof.py > /dev/null
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE -d $O/m4 -o m -- python3 $R/tools/fk_prof.py > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/rd -o rd -- python3 $R/tools/fk_prof.py > /dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/wr -o wr -- python3 $R/tools/fk_prof.py > /dev/null
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE -d $O/m3 -o m -- python3 $R/tools/fk_prof.py > /dev/null
cd $R
TAG=${1:-r05}
{
  echo "# joints-only FK (k_fk_joints_dma) at 2^20 poses, counters per launch ($TAG; tools/fk_pmc.sh)"
  echo
  python3 tools/rocpd_summary.py $(find $O/m2 -name "*.db") | grep "k_fk\|^| kernel\|^|---"
  echo
  python3 tools/rocpd_summary.py --counters $(find $O/m2 -name "*.db") | grep "k_fk\|^| kernel\|^|---"
  echo
  python3 tools/rocpd_summary.py --counters $(find $O/m4 -name "*.db") | grep "k_fk\|^| kernel\|^|---"
  echo
  python3 tools/rocpd_summary.py --pmc $(find $O/rd $O/wr -name "*.db") | grep "k_fk\|^| kernel\|^|---"
  echo
  python3 tools/rocpd_summary.py --counters $(find $O/m3 -name "*.db") | grep "k_fk\|^| kernel\|^|---"
} > gpurun_out/${TAG}_fk_pmc.md
rm -rf $O
cat gpurun_out/${TAG}_fk_pmc.md
