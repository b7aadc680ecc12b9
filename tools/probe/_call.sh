cd $GRAFT_REPO_ROOT
python tools/perf_quick.py --games chaser --check 256x400 2>&1 | tail -1
for rep in 1 2; do
python tools/perf_quick.py --games chaser --check 0x0 --settle 600 --steps 256 2>&1 | tail -1 | sed "s/^/rgb base: /"
python tools/perf_quick.py --games chaser --check 0x0 --settle 600 --steps 256 --lib procgen2_amd/lib/libpg_exp_words.so 2>&1 | tail -1 | sed "s/^/word base: /"
done
timeout 1500 python -m pytest tests -m gpu -x -q -k "chaser or snapshot or mixed or modes" 2>&1 | tail -3
