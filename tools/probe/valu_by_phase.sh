#!/bin/bash
# dynamic instruction counts of coinrun's render kernel with parts of it switched off (ablation build)
for f in 0 128 2 8 130 138 14; do
  echo "== debug $f"
  PG_DEBUG=$f PG_LIB=procgen2_amd/lib/libprocgen2_hip_ablate.so PG_SETTLE=200 bash tools/pmc_quick.sh r02i/abl_$f "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" 2>/dev/null | grep SQ_
done
