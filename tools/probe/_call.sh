cd $GRAFT_REPO_ROOT
python tools/perf_quick.py --games coinrun --check 256x300 2>&1 | tail -1
for rep in 1 2; do
python tools/perf_quick.py --games coinrun --check 0x0 --settle 600 --steps 256 2>&1 | tail -1 | sed "s/^/sure 24: /"
for v in sure0 sure16 sure64; do python tools/perf_quick.py --games coinrun --check 0x0 --settle 600 --steps 256 --lib procgen2_amd/lib/libpg_exp_$v.so 2>&1 | tail -1 | sed "s/^/$v: /"; done
done
