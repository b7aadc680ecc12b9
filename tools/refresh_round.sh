R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
tools/refresh_all.sh r04_final 2>&1 | tail -12
tools/kernel_stats_all.sh r04_k 2>&1 | tail -8
tools/pmc_all_games.sh r04_s 2>&1 | tail -12
( cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16 && rm -rf /tmp/mx && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mx -- python3 $R/bench.py --workload mixed --steps 128 --warmup 16 > $R/gpurun_out/r04_m_mixed.log 2>&1; python3 $R/tools/mixed_timeline.py $(ls /tmp/mx/*/*kernel_trace.csv | head -1) $R/gpurun_out/r04_m_mixed_kernel_timeline.json | tail -5; cp $(ls /tmp/mx/*/*kernel_stats.csv | head -1) $R/gpurun_out/r04_m_mixed_kernel_stats.csv )
