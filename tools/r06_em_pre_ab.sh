#!/bin/bash
# Run on the GPU box:  bash tools/r06_em_pre_ab.sh -- the fused post_dense + Euler-Maruyama launch of the sampler at 500 samples (cfg 3), interleaved:
#   before   tools/bin/libdposer_hip_pre_em.so   state loads and Philox normals behind the last MFMA, 64x32 tiling on 4 slots x 2 k-blocks
#   pre      tools/bin/libdposer_hip_pre_kb2.so  loads / normals in front of the K loop (gemm.h EpiPre), same pipeline
#   shipped  dposer_amd/libdposer_hip.so         the same + 3 slots x 4 k-blocks
cd "$(dirname "$0")/.."
for r in 1 2 3; do
  for v in before:tools/bin/libdposer_hip_pre_em.so pre:tools/bin/libdposer_hip_pre_kb2.so shipped:dposer_amd/libdposer_hip.so; do
    echo "${v%%:*} (run $r): $(DPOSER_LIB_PATH=$PWD/${v#*:} python tools/config_timings.py cfg3 2>/dev/null | grep cfg3)"
  done
done
