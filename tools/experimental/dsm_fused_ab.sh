#!/bin/bash
# A/B on one box: post_dense + DSM loss as one launch (DPOSER_DSM_FUSED=1, opt-in) vs GEMM -> res -> k_dsm (DPOSER_DSM_FUSED=0, the default); ms per training step
COMMON="--no-extra --no-cpu-baseline --no-live-roofline --steps 200 --warmup 20"
for B in 1280 8192 65536; do
  for F in 0 1 0 1; do
    P=$(DPOSER_DSM_FUSED=$F python3 bench.py --global-batch $B $COMMON 2>/dev/null | grep '^{' | python3 -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "B=$B DPOSER_DSM_FUSED=$F  $P ms"
  done
done
