cd $GRAFT_REPO_ROOT
python tools/perf_quick.py --games coinrun --check 256x300 2>&1 | tail -1
for rep in 1 2 3; do
python tools/perf_quick.py --games coinrun --check 0x0 --settle 600 --steps 256 2>&1 | tail -1
python tools/perf_quick.py --games coinrun --check 0x0 --settle 600 --steps 256 --lib procgen2_amd/lib/libpg_exp_before.so 2>&1 | tail -1 | sed 's/^/   before: /'
done
timeout 900 python -m pytest tests -m gpu -x -q -k "coinrun" 2>&1 | tail -3
