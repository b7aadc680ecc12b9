#!/bin/bash
# usage: tools/refresh_round.sh rNN      (GPU box; run tools/pmc_traffic.sh rNN_final FIRST, in a call of its own, and copy
# its gpurun_out/rNN_final_traffic.json to profiles/rNN_final_traffic_pmc.json: the bench lines quote it as roofline.traffic)
# The whole round-end set: bench lines of every game + mixed, per-game kernel stats, render counters, the mixed trace.
R=${GRAFT_REPO_ROOT:-/root/repo}
RN=${1:-r05}
cd $R
tools/refresh_all.sh ${RN}_final 2>&1 | tail -12
tools/kernel_stats_all.sh ${RN}_k 2>&1 | tail -8
tools/pmc_all_games.sh ${RN}_s 2>&1 | tail -12
( cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16 && rm -rf /tmp/mx && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mx -- python3 $R/bench.py --workload mixed --steps 128 --warmup 16 --no-cpu-baseline > $R/gpurun_out/${RN}_m_mixed.log 2>&1; python3 $R/tools/mixed_timeline.py $(ls /tmp/mx/*/*kernel_trace.csv | head -1) $R/gpurun_out/${RN}_m_mixed_kernel_timeline.json | tail -5; cp $(ls /tmp/mx/*/*kernel_stats.csv | head -1) $R/gpurun_out/${RN}_m_mixed_kernel_stats.csv )
