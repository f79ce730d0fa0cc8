#!/bin/bash
# A/B on one box: time-branch dgrad as one k-split per layer + reduce pass (default up to 2048 samples) vs the one launch; ms per training step
COMMON="--no-extra --no-cpu-baseline --no-live-roofline --steps 300 --warmup 30"
for B in 1280 2048 4096; do
  for F in 0 8192 0 8192; do
    P=$(DPOSER_SILU_SPLIT_MAX=$F python3 bench.py --global-batch $B $COMMON 2>/dev/null | grep '^{' | python3 -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "B=$B DPOSER_SILU_SPLIT_MAX=$F  $P ms"
  done
done
