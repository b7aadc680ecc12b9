cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "step_many" 2>&1 | tail -3
python __graft_entry__.py smoke 2>&1 | tail -2
bash tools/refresh_round.sh r05 2>&1 | tail -40
