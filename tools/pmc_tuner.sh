#!/bin/bash
# Run on the GPU box:  bash tools/pmc_tuner.sh <tuner binary in tools/bin> <K>   -- matrix-pipe / LDS counters of the TUNE_R3 kernels
# usage: bash tools/bin/pmc_tuner.sh <binary> <K>
B=$1; K=${2:-4096}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export TUNE_R3=1 TUNE_K=$K
O=$R/gpurun_out/tpmc; rm -rf $O
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $O/t1 -o t -- $R/tools/bin/$B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $O/t2 -o t -- $R/tools/bin/$B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $O/t3 -o t -- $R/tools/bin/$B > /dev/null 2>&1
cd $R
python3 tools/rocpd_summary.py --mfma $(find $O/t1 $O/t2 -name "*.db") | grep "gemm_ft\|^| kernel"
python3 tools/rocpd_summary.py --counters $(find $O/t3 -name "*.db") | grep "gemm_ft\|^| kernel"
rm -rf $O
