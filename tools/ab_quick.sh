#!/bin/bash
# usage: tools/ab_quick.sh [game ...]        (GPU box)
# Same-box A/B of the closing engine of the round before (procgen2_amd/lib_ref/libprocgen2_hip_r05.so, built from that
# round's last commit: `git worktree add /tmp/r05 <commit>; python -m procgen2_amd.build` there) against the tree's own:
# tools/perf_quick.py per game, one process each, 65 536 envs, no lock-step.  Boxes of the pool differ by 1-3 %; a pair
# taken seconds apart on one box does not.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for G in ${@:-coinrun maze bossfight climber caveflyer chaser jumper}; do
  for L in procgen2_amd/lib_ref/libprocgen2_hip_r05.so ""; do
    tag=$([ -n "$L" ] && echo "r05" || echo "now")
    out=$(timeout 120 python tools/perf_quick.py --games $G --check 0x0 ${L:+--lib $L} 2>/dev/null | tail -1)
    echo "$tag $out"
  done
done
