cd $GRAFT_REPO_ROOT
bash tools/bound_table.sh r05 2>&1 | tail -5
python tools/bound_table.py gpurun_out/r05_bounds_raw.json > gpurun_out/r05_bounds.md
tail -5 gpurun_out/r05_bounds.md
