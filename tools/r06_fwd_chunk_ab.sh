#!/bin/bash
# LBS forward in chunks on ONE stream (DPOSER_LBS_FWD_CHUNK=n DPOSER_LBS_FWD_CHUNK_SERIAL=1): does a chunk's offsets stay in the memory-side cache?
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for rep in 1 2; do
  for c in 0 256 512 1024 2048; do
    echo "## chunk $c serial (run $rep): $(DPOSER_LBS_FWD_CHUNK=$c DPOSER_LBS_FWD_CHUNK_SERIAL=1 python3 tools/lbs_ab.py child 2>&1 | grep 'n=  4096\|n= 16384' | tr '\n' ' ')"
  done
done
