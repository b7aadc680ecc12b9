#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py for every game (default modes): per-kernel average durations.
# --settle 3072: all envs are made in the same step and the first few hundred steps after that are a transient (episodes in
# phase with each other); with 3 488 launches per kernel the averages are steady state to within a few per cent.
# Output: gpurun_out/<tag>_<game>_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04_k}
mkdir -p $(dirname $R/gpurun_out/${TAG}_x)
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16   # the profiler initialises HIP before the library can: same configuration as the bench line
for G in coinrun maze bossfight climber caveflyer chaser jumper; do
  rm -rf /tmp/ks_$G
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$G -- python3 $R/bench.py --game $G --settle 3072 --steps 128 --warmup 32 --no-cpu-baseline > $R/gpurun_out/${TAG}_$G.log 2>&1
  f=$(ls /tmp/ks_$G/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" $R/gpurun_out/${TAG}_${G}_kernel_stats.csv
done
ls -la $R/gpurun_out/ | grep kernel_stats
