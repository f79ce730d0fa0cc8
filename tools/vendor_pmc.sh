#!/bin/bash
# Run on the GPU box:  bash tools/vendor_pmc.sh [K]
# Matrix-pipe counters (MFMA busy cycles, effective clock, issue stalls) of the vendor library's GEMM (torch.matmul -> hipBLASLt) and of
# this library's plain 256x256 ring loop on the same 65536 x 1024 x K problem: which of the two -- cycles or clock -- is the vendor
# kernel's advantage at long K?   -> gpurun_out/vendor_pmc_K<K>.md
K=${1:-4096}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
export VENDOR_K=$K TUNE_R3=1 TUNE_K=$K
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
P3="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE"
O=$R/gpurun_out/vpmc
rm -rf $O
rocprofv3 --kernel-trace --pmc $P1 -d $O/v1 -o v -- python3 $R/tools/vendor_gemm_reference.py > $R/gpurun_out/vendor_pmc_K$K.log 2>&1
rocprofv3 --kernel-trace --pmc $P2 -d $O/v2 -o v -- python3 $R/tools/vendor_gemm_reference.py >> $R/gpurun_out/vendor_pmc_K$K.log 2>&1
rocprofv3 --kernel-trace --pmc $P3 -d $O/v3 -o v -- python3 $R/tools/vendor_gemm_reference.py >> $R/gpurun_out/vendor_pmc_K$K.log 2>&1
rocprofv3 --kernel-trace --pmc $P1 -d $O/t1 -o t -- $R/tools/bin/tune_gemm >> $R/gpurun_out/vendor_pmc_K$K.log 2>&1
rocprofv3 --kernel-trace --pmc $P2 -d $O/t2 -o t -- $R/tools/bin/tune_gemm >> $R/gpurun_out/vendor_pmc_K$K.log 2>&1
rocprofv3 --kernel-trace --pmc $P3 -d $O/t3 -o t -- $R/tools/bin/tune_gemm >> $R/gpurun_out/vendor_pmc_K$K.log 2>&1
cd $R
{ echo "## K = $K: vendor (torch.matmul)"; python3 tools/rocpd_summary.py --mfma $(find $O/v1 $O/v2 -name "*.db");
  echo; python3 tools/rocpd_summary.py --counters $(find $O/v3 -name "*.db");
  echo; echo "## K = $K: this library (tools/bin/tune_gemm TUNE_R3)"; python3 tools/rocpd_summary.py --mfma $(find $O/t1 $O/t2 -name "*.db");
  echo; python3 tools/rocpd_summary.py --counters $(find $O/t3 -name "*.db"); } > gpurun_out/vendor_pmc_K$K.md
rm -rf $O
cat gpurun_out/vendor_pmc_K$K.md
