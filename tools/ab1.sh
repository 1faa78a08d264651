#!/bin/bash
# one pass A/B of library builds: ab1.sh "<T list>" lib1 lib2 ... (development aid; ab.sh runs two passes)
Ts="$1"; shift
for T in $Ts; do for L in "$@"; do
  r=$(LBM_HIP_LIB=$PWD/$L timeout -k 10 120 python tools/quick_bench.py 16384 16384 ${STEPS:-16} ${BOUNDARY:-periodic} $T 0 6 ${DTYPE:-f32} ${OMEGA:-1e-14} ${MASK:-none} 2>/dev/null | grep GLUPS | tail -1 | awk '{print $9}') || exit 1
  echo "T=$T $L $r"
done; done
