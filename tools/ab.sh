#!/bin/bash
# A/B of library builds on one box: ab.sh "<T list>" lib1 lib2 ... (development aid)
# Stops at the first run that fails or prints no rate: after a GPU fault nothing more may be started.
set -o pipefail
Ts="$1"; shift
for rep in 1 2; do for T in $Ts; do for L in "$@"; do
  r=$(LBM_HIP_LIB=$PWD/$L timeout -k 10 120 python tools/quick_bench.py 16384 16384 20 ${BOUNDARY:-periodic} $T 0 6 ${DTYPE:-f32} ${OMEGA:-1e-14} ${MASK:-none} 2>/dev/null | grep GLUPS | tail -1 | awk '{print $9}') || { echo "T=$T $L FAILED"; exit 1; }
  [ -n "$r" ] || { echo "T=$T $L no result"; exit 1; }
  echo "T=$T $L $r"
done; done; done
