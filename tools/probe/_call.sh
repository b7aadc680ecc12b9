cd $GRAFT_REPO_ROOT
for g in maze climber caveflyer jumper coinrun; do
for rep in 1 2; do
python tools/perf_quick.py --games $g --check 0x0 --settle 600 --steps 256 2>&1 | tail -1
PG_SEPARATE_INSTALL=1 python tools/perf_quick.py --games $g --check 0x0 --settle 600 --steps 256 2>&1 | tail -1 | sed 's/^/   separate: /'
done; done
echo "== mixed: default side-stream priority, then low"
python bench.py --workload mixed --no-cpu-baseline --steps 256 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mixed', d['value']/1e6)"
PG_SIDE_PRIORITY=low python bench.py --workload mixed --no-cpu-baseline --steps 256 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mixed low', d['value']/1e6)"
python bench.py --workload mixed --no-cpu-baseline --steps 256 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mixed', d['value']/1e6)"
PG_SIDE_PRIORITY=low python bench.py --workload mixed --no-cpu-baseline --steps 256 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mixed low', d['value']/1e6)"
