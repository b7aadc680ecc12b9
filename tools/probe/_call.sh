cd $GRAFT_REPO_ROOT
for g in coinrun bossfight chaser jumper; do echo "== $g"; python tools/probe/wave_timeline.py $g procgen2_amd/lib/libpg_exp_tl_$g.so 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r05_wave_timelines.txt
