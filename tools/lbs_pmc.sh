#!/bin/bash
# Run on the GPU box:  bash tools/lbs_pmc.sh   -- rocprofv3 counter passes (SQ wait / LDS / FETCH / WRITE) of four LBS forward + backward iterations at 4096 poses
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
[ -f "$R/bench.py" ] || { echo "$0: $R is not the repository root" >&2; exit 2; }
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/lbspmc; rm -rf $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $O/m2 -o m -- python3 $R/tools/lbs_prof.py > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/rd -o rd -- python3 $R/tools/lbs_prof.py > /dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/wr -o wr -- python3 $R/tools/lbs_prof.py > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/frd -o rd -- python3 $R/tools/lbs_prof.py fwd > /dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/fwr -o wr -- python3 $R/tools/lbs_prof.py fwd > /dev/null
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE -d $O/m3 -o m -- python3 $R/tools/lbs_prof.py > /dev/null
cd $R
TAG=${1:-r04}
{
  echo "# LBS forward + backward at 4096 poses (tools/lbs_prof.py: four iterations), counters per launch ($TAG, commit $(cat $R/.head_rev 2>/dev/null || echo snapshot))"
  echo
  python3 tools/rocpd_summary.py --mfma $(find $O/m2 -name "*.db") | grep "k_skin\|^| kernel\|^|---\|k_fk\|gemm\|k_split\|k_extra\|k_sum"
  echo
  python3 tools/rocpd_summary.py --pmc $(find $O/rd $O/wr -name "*.db") | grep "k_skin\|^| kernel\|^|---\|k_fk\|gemm\|k_split\|k_extra\|k_sum"
  echo
  python3 tools/rocpd_summary.py --counters $(find $O/m3 -name "*.db") | grep "k_skin\|^| kernel\|^|---"
} > gpurun_out/${TAG}_lbs_pmc.md
python3 tools/rocpd_summary.py --pmc-json $(find $O/rd $O/wr -name "*.db") > gpurun_out/${TAG}_lbs_pmc_traffic_raw.json
python3 tools/rocpd_summary.py --pmc-json $(find $O/frd $O/fwr -name "*.db") > gpurun_out/${TAG}_lbs_pmc_traffic_fwd_raw.json
# bytes of one forward and of one forward + backward call (per-launch averages x launches per call) -> profiles/pmc_lbs_traffic.json
python3 - <<PY
import json
t = json.load(open("gpurun_out/${TAG}_lbs_pmc_traffic_raw.json"))
tf = json.load(open("gpurun_out/${TAG}_lbs_pmc_traffic_fwd_raw.json"))
iters = 4
lib = lambda k: not k.startswith("at::") and "rocclr" not in k
tot_f = sum((v["read_MB"] + v["write_MB"]) * v["launches"] for k, v in tf.items() if lib(k))      # forward-only passes (round 6: measured by itself)
tot_all = sum((v["read_MB"] + v["write_MB"]) * v["launches"] for k, v in t.items() if lib(k))
out = {"fwd_bytes": tot_f / iters * 1e6, "fwd_bwd_bytes": tot_all / iters * 1e6, "poses": 4096,
       "source": "tools/lbs_pmc.sh: FETCH_SIZE (x2, gfx950) / WRITE_SIZE passes of tools/lbs_prof.py (forward + backward) and of tools/lbs_prof.py fwd (forward alone)"}
json.dump(out, open("gpurun_out/${TAG}_pmc_lbs_traffic.json", "w"), indent=1)
print(out)
PY
rm -rf $O
cat gpurun_out/${TAG}_lbs_pmc.md
