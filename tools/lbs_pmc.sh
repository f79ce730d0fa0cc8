#!/bin/bash
# Run on the GPU box:  bash tools/lbs_pmc.sh   -- rocprofv3 counter passes (SQ wait / LDS / FETCH / WRITE) of four LBS forward + backward iterations at 4096 poses
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/lbspmc; rm -rf $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $O/m2 -o m -- python3 $R/tools/lbs_prof.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/rd -o rd -- python3 $R/tools/lbs_prof.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/wr -o wr -- python3 $R/tools/lbs_prof.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE -d $O/m3 -o m -- python3 $R/tools/lbs_prof.py > /dev/null 2>&1
cd $R
python3 tools/rocpd_summary.py --mfma $(find $O/m2 -name "*.db") | grep "k_skin\|^| kernel\|k_fk\|gemm"
python3 tools/rocpd_summary.py --pmc $(find $O/rd $O/wr -name "*.db") | grep "k_skin\|^| kernel"
python3 tools/rocpd_summary.py --counters $(find $O/m3 -name "*.db") | grep "k_skin\|^| kernel"
rm -rf $O
