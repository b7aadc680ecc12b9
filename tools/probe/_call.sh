cd $GRAFT_REPO_ROOT
bash tools/probe/chaser_logic_phases.sh run r05m/chl 2>&1 | tee gpurun_out/r05m_chaser_logic_phases.txt
python tools/perf_quick.py --games bossfight,caveflyer,climber,jumper,chaser --check 128x150 2>&1 | tail -5
