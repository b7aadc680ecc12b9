cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05c
python tools/perf_quick.py --games coinrun --check 256x200 2>&1 | tail -3
timeout 900 python -m pytest tests -m gpu -x -q -k "coinrun or cadence or step_times or abi or symbols" 2>&1 | tail -8
bash tools/kstats_quick.sh r05c/trim coinrun 2>&1 | grep "coinrun::"
python bench.py --steps 256 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], json.dumps(d['roofline'], indent=0)[:1800])"
