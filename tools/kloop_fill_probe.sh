#!/bin/bash
# Issue-slot probe of the shipped asm K-loop stage (tools/kloop_fill_probe.hip).
#   tools/kloop_fill_probe.sh build     cross-compiles every variant into tools/bin/kfill/ (no GPU needed)
#   tools/kloop_fill_probe.sh run [R]   runs them, R interleaved rounds (default 2), on the GPU box
set -e
cd "$(dirname "$0")/.."
OUT=tools/bin/kfill
VARIANTS="0,0 1,0 2,0 3,0 4,0 5,0 6,0 0,1 0,2 2,1 3,1 4,1 3,2 4,2 6,2"
if [ "$1" = build ]; then
  mkdir -p $OUT
  for v in $VARIANTS; do
    n=${v%,*}; m=${v#*,}
    python3 tools/gen_kloop_fill.py $n $m > $OUT/fill_n${n}_s${m}.inc
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Idposer_amd/csrc -Iinclude -Itools -DDPOSER_KLOOP_FILL_INC="\"$PWD/$OUT/fill_n${n}_s${m}.inc\"" \
        tools/kloop_fill_probe.hip -o $OUT/kfill_n${n}_s${m} &
    if (( $(jobs -r | wc -l) >= 8 )); then wait -n; fi
  done
  # the same VALU count as packed fp32 math / as plain fma only
  for n in 2 3 4; do
    python3 tools/gen_kloop_fill.py $n 0 --pk > $OUT/fill_n${n}_pk.inc
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Idposer_amd/csrc -Iinclude -Itools -DFILL_TAG='"pk"' -DDPOSER_KLOOP_FILL_INC="\"$PWD/$OUT/fill_n${n}_pk.inc\"" \
        tools/kloop_fill_probe.hip -o $OUT/kfill_n${n}_pk &
    python3 tools/gen_kloop_fill.py $n 0 --plain > $OUT/fill_n${n}_plain.inc
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Idposer_amd/csrc -Iinclude -Itools -DFILL_TAG='"fma"' -DDPOSER_KLOOP_FILL_INC="\"$PWD/$OUT/fill_n${n}_plain.inc\"" \
        tools/kloop_fill_probe.hip -o $OUT/kfill_n${n}_plain &
    if (( $(jobs -r | wc -l) >= 8 )); then wait -n; fi
  done
  # store placement / cache-policy variants of one tile store per stage (the volume of a training epilogue's output tiles)
  for tag in late "late --nt" "late --sc1" nt; do
    name=$(echo $tag | tr -d ' -')
    python3 tools/gen_kloop_fill.py 0 1 --$tag > $OUT/fill_n0_s1_$name.inc
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Idposer_amd/csrc -Iinclude -Itools -DFILL_TAG="\"$name\"" -DDPOSER_KLOOP_FILL_INC="\"$PWD/$OUT/fill_n0_s1_$name.inc\"" \
        tools/kloop_fill_probe.hip -o $OUT/kfill_n0_s1_$name &
  done
  wait
  ls $OUT | grep -v inc
  exit 0
fi
R=${2:-2}
for r in $(seq 1 $R); do
  for f in $(ls $OUT | grep -v '\.inc$' | grep "${3:-kfill}" | sort -V); do timeout 120 $OUT/$f 5; done
done
