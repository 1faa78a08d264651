#!/bin/bash
# small-grid A/B: small.sh lib... ; the reference's own scene sizes, reference edges
for L in "$@"; do for cfg in "400 800 200 reference 6 0 6" "400 800 200 reference 3 0 3" "400 800 200 reference 4 0 4" "1355 3782 60 reference 6 0 6" "1024 1024 100 reference 6 0 6" "2160 3840 40 reference 6 0 6"; do
  r=$(LBM_HIP_LIB=$PWD/$L timeout -k 10 60 python tools/quick_bench.py $cfg 2>/dev/null | grep GLUPS | tail -1 | awk '{print $8, $9}') || exit 1
  echo "$L [$cfg] $r"
done; done
