#!/bin/bash
# scratch: A/B two builds of one game on the GPU box.  usage: ab.sh GAME "FLAGS_A" "FLAGS_B"
for v in "$2" "$3" "$2" "$3"; do
  touch procgen2_amd/csrc/pg_render.h
  PG_MORE_FLAGS="$v" python -m procgen2_amd.build --quiet > /dev/null 2>&1
  echo "=== [$v]"; timeout 200 python tools/perf_quick.py --games $1 --check 0x0 2>&1 | grep -v amdgpu | tail -1
done
