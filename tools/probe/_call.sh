cd $GRAFT_REPO_ROOT
python tools/perf_quick.py --games chaser --check 256x400 --settle 600 --steps 256 2>&1 | tail -1
python tools/perf_quick.py --games chaser --check 0x0 --settle 600 --steps 256 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -x -q -k "chaser or snapshot or mixed or modes or levels" 2>&1 | tail -3
