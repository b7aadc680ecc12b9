cd $GRAFT_REPO_ROOT
python tools/perf_quick.py --games caveflyer,climber --check 128x200 2>&1 | tail -2
for g in caveflyer climber; do for rep in 1 2; do
python tools/perf_quick.py --games $g --check 0x0 --settle 600 --steps 256 2>&1 | tail -1
PG_SEPARATE_INSTALL=1 python tools/perf_quick.py --games $g --check 0x0 --settle 600 --steps 256 2>&1 | tail -1 | sed 's/^/   separate: /'
done; done
bash tools/kstats_quick.sh r05p/cave caveflyer 2>&1 | grep "caveflyer::"
