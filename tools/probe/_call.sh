cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05k
python tools/perf_quick.py --games coinrun --check 256x300 2>&1 | tail -2
bash tools/kstats_quick.sh r05k/fused coinrun 2>&1 | grep -E "coinrun::"
timeout 900 python -m pytest tests -m gpu -x -q -k "coinrun" 2>&1 | tail -6
