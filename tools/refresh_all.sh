#!/bin/bash
# Round-end refresh on the GPU box: bench lines of every game + mixed, rocprof kernel stats and PMC traffic of the
# headline run.  Everything lands in gpurun_out/<tag>_*.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04_final}
cd $R
mkdir -p $(dirname gpurun_out/${TAG}_x)
python bench.py --game coinrun 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_coinrun.json
for G in maze bossfight climber caveflyer chaser jumper; do
  timeout 300 python bench.py --game $G 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_$G.json   # (with its CPU baseline: ≈ 20 s of the oracle on the host cores)
done
timeout 300 python bench.py --workload mixed 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_mixed.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_coinrun_driver_args.json
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks_final && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_final -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/${TAG}_rocprof.log 2>&1; cp $(ls /tmp/ks_final/*/*kernel_stats.csv | head -1) $R/gpurun_out/${TAG}_coinrun_kernel_stats.csv )
# (tools/pmc_traffic.sh runs BEFORE this script, in a call of its own: its result, copied to profiles/*traffic_pmc.json, is
# what the bench lines above quote as roofline.traffic — stamped with the fingerprint of the kernel sources it was taken on)
for f in gpurun_out/${TAG}_bench_*.json; do python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); c=d["config"]
print(c.get("game","mixed"), round(d["value"]/1e6,1), "M env-steps/s", round(d["ms_per_step"],3), "ms", "roofline", round(d.get("roofline",{}).get("frac",0),3))
PY
done
head -4 gpurun_out/${TAG}_coinrun_kernel_stats.csv | cut -c1-200
# the regression gate: every line against the committed line of the same workload of the round before (tools/check_bench.py)
AGAINST=${CHECK_AGAINST:-profiles/bench_r05}
for f in gpurun_out/${TAG}_bench_*.json; do
  case $f in *driver_args*) continue;; esac
  python tools/check_bench.py "$f" --against $AGAINST || echo "check_bench: $f: exit $?"
done
