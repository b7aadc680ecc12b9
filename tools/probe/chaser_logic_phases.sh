#!/bin/bash
# usage (build container): tools/probe/chaser_logic_phases.sh build     → experiment libraries lib/libpg_exp_chl<bits>.so
#       (GPU box):         tools/probe/chaser_logic_phases.sh run TAG   → time and vector instructions per wave of logic_kernel per build
# where chaser's logic kernel's instructions are.  PG_CHASER_SKIP bits leave a part out (the rollout then
# differs — agents stand still, enemies do not move —, so this is an inventory, not an A/B): 1 agent, 2 enemies, 4 points, 8 draw list.
R=${GRAFT_REPO_ROOT:-/root/repo}
if [ "$1" = build ]; then
  for b in 1 2 4 8 15; do python3 $R/tools/build_exp.py chaser chl$b -DPG_CHASER_SKIP=$b | tail -1; done
  exit 0
fi
TAG=$2
mkdir -p $R/gpurun_out/$(dirname $TAG)
for b in 0 1 2 4 8 15; do
  lib=$R/procgen2_amd/lib/libpg_exp_chl$b.so; [ $b = 0 ] && lib=$R/procgen2_amd/lib/libprocgen2_hip.so
  echo "== skip $b"
  PROCGEN2_HIP_LIB=$lib bash $R/tools/kstats_quick.sh ${TAG}_skip$b chaser 2>&1 | grep -E "logic_kernel" | cut -c1-110
  PG_LIB=${lib#$R/} PG_GAME=chaser PG_KERNEL=logic_kernel bash $R/tools/pmc_quick.sh ${TAG}_skip$b "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" 2>&1 | grep "per wave"
done
