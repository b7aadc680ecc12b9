cd $GRAFT_REPO_ROOT
echo "== chaser gang variants (product: gang 8, waves 4)"
python tools/perf_quick.py --games chaser --check 0x0 --settle 600 --steps 256 2>&1 | tail -1
for v in gang4 gang16 gang4w2 gang4w8; do python tools/perf_quick.py --games chaser --check 64x300 --settle 600 --steps 256 --lib procgen2_amd/lib/libpg_exp_$v.so 2>&1 | tail -1 | sed "s/^/  $v: /"; done
python tools/perf_quick.py --games chaser --check 0x0 --settle 600 --steps 256 2>&1 | tail -1
echo "== mixed: default, then slow games on high-priority streams"
m() { python bench.py --workload mixed --no-cpu-baseline --steps 256 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e6,2))"; }
m default
PG_STREAM_PRIORITY_JUMPER=-1 PG_STREAM_PRIORITY_CHASER=-1 PG_STREAM_PRIORITY_BOSSFIGHT=-1 m "jumper+chaser+bossfight high"
PG_STREAM_PRIORITY_JUMPER=-1 m "jumper high"
m default
PG_STREAM_PRIORITY_JUMPER=-1 PG_STREAM_PRIORITY_CHASER=-1 PG_STREAM_PRIORITY_BOSSFIGHT=-1 PG_STREAM_PRIORITY_CAVEFLYER=-1 m "four slow high"
