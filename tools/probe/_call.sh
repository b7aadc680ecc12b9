cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -k "chaser or snapshot or mixed or modes or levels or gym or human_frame" 2>&1 | tail -4
for rep in 1 2; do python tools/perf_quick.py --games chaser --check 0x0 --settle 600 --steps 256 2>&1 | tail -1; done
