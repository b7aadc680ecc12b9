#!/bin/bash
# timing experiment: composer batch size (rows per memory round trip)
for B in 4 8 16; do
  sed -i "s/^#define PG_BATCH .*/#define PG_BATCH $B/" procgen2_amd/csrc/pg_render.h
  python3 -m procgen2_amd.build --quiet >/dev/null 2>&1
  echo "== batch $B"; python tools/ablate_render.py | head -1
done
