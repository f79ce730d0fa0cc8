#!/bin/bash
# Run on the GPU box:  bash tools/profile_bench.sh <tag>     (e.g. r01)
# 1. rocprofv3 kernel trace of the default bench.py command            -> gpurun_out/prof_<tag>/
# 2. separate PMC passes (FETCH_SIZE, WRITE_SIZE) of a short bench run  -> gpurun_out/pmc_<tag>_{rd,wr}/
# 3. markdown / json summaries                                          -> gpurun_out/<tag>_*.md|json  (copy into profiles/)
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$TAG $R/gpurun_out/pmc_${TAG}_rd $R/gpurun_out/pmc_${TAG}_wr
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o bench -- python3 $R/bench.py > $R/gpurun_out/${TAG}_bench_default.log 2>&1
grep '^{' $R/gpurun_out/${TAG}_bench_default.log | tail -1 > $R/gpurun_out/${TAG}_bench_default.json
SHORT="--steps 3 --warmup 1 --no-cpu-baseline --sampler-steps 20"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_${TAG}_rd -o rd -- python3 $R/bench.py $SHORT > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_${TAG}_wr -o wr -- python3 $R/bench.py $SHORT > /dev/null 2>&1
# 4. matrix-pipe evidence: MFMA busy cycles, effective clock (GRBM_GUI_ACTIVE / duration), issue stalls -- two SQ passes + GRBM
rm -rf $R/gpurun_out/pmc_${TAG}_m1 $R/gpurun_out/pmc_${TAG}_m2
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_${TAG}_m1 -o m1 -- python3 $R/bench.py $SHORT > $R/gpurun_out/${TAG}_pmc_m1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_${TAG}_m2 -o m2 -- python3 $R/bench.py $SHORT > $R/gpurun_out/${TAG}_pmc_m2.log 2>&1
cd $R
HEADREV=$(git -C $R rev-parse --short HEAD 2>/dev/null || cat $R/.head_rev 2>/dev/null || echo snapshot)
python3 tools/rocpd_summary.py --mfma $(find gpurun_out/pmc_${TAG}_m1 gpurun_out/pmc_${TAG}_m2 -name "*.db") > gpurun_out/${TAG}_pmc_mfma.md
python3 tools/rocpd_summary.py --mfma-json $(find gpurun_out/pmc_${TAG}_m1 gpurun_out/pmc_${TAG}_m2 -name "*.db") > gpurun_out/${TAG}_pmc_mfma.json   # -> profiles/pmc_mfma.json (bench.py looks it up)
# per-dispatch view of the two epilogue-heavy training GEMMs (position 0 of EpiGNBwd = the K = 64 post_dense launch: epilogue only;
# position 0 of EpiGN<train> = layer 0 with K = 576 against K = 1536): VALU / MFMA instructions, busy cycles and durations per position
{
  echo "# per-dispatch counters of the GroupNorm training GEMMs, folded over the steps ($TAG)"
  python3 tools/rocpd_summary.py --dispatches "256x256,EpiGNBwd" 5 $(find gpurun_out/pmc_${TAG}_m1 gpurun_out/pmc_${TAG}_m2 -name "*.db")
  python3 tools/rocpd_summary.py --dispatches "256x256,EpiGN<train>" 5 $(find gpurun_out/pmc_${TAG}_m1 gpurun_out/pmc_${TAG}_m2 -name "*.db")
} > gpurun_out/${TAG}_epilogue_dispatches.md 2>&1
rm -rf gpurun_out/pmc_${TAG}_m1 gpurun_out/pmc_${TAG}_m2
python3 tools/rocpd_summary.py $(find gpurun_out/prof_$TAG -name "*.db" | head -1) > gpurun_out/${TAG}_bench_kernel_stats.md
python3 tools/rocpd_summary.py --pmc $(find gpurun_out/pmc_${TAG}_rd gpurun_out/pmc_${TAG}_wr -name "*.db") > gpurun_out/${TAG}_pmc_hbm_traffic.md
python3 tools/rocpd_summary.py --pmc-json $(find gpurun_out/pmc_${TAG}_rd gpurun_out/pmc_${TAG}_wr -name "*.db") > gpurun_out/${TAG}_pmc_hbm_traffic.json
# the raw rocpd databases can exceed what gpurun copies back (64 MiB): keep the summaries only
rm -rf gpurun_out/prof_$TAG gpurun_out/pmc_${TAG}_rd gpurun_out/pmc_${TAG}_wr
for f in gpurun_out/${TAG}_*.md; do sed -i "1s/^/commit $HEADREV (tools\/profile_bench.sh $TAG)\n\n/" $f; done
head -40 gpurun_out/${TAG}_epilogue_dispatches.md; head -30 gpurun_out/${TAG}_bench_kernel_stats.md; head -30 gpurun_out/${TAG}_pmc_mfma.md; head -24 gpurun_out/${TAG}_pmc_hbm_traffic.md; cat gpurun_out/${TAG}_bench_default.json
