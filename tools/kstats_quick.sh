#!/bin/bash
# usage: tools/kstats_quick.sh TAG GAME [bench.py args…]
# rocprofv3 --kernel-trace --stats of a short bench.py run of one game; prints the per-kernel table (calls, average ns)
# and keeps the csv as gpurun_out/TAG_GAME_kernel_stats.csv.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; G=$2; shift 2
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16   # the profiler initialises HIP before the library can: same configuration as the bench line
rm -rf /tmp/kq_$G
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kq_$G -- python3 $R/bench.py --game $G --settle 512 --steps 128 --warmup 16 --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_$G.log 2>&1
f=$(ls /tmp/kq_$G/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" $R/gpurun_out/${TAG}_${G}_kernel_stats.csv && python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Name"].split("(")[0].split("::")[-2:] if "::" in r["Name"] else [r["Name"][:60]]
    print("%-40s calls %6s  avg %10.1f us  total %6.2f %%" % ("::".join(name)[:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
tail -2 $R/gpurun_out/${TAG}_$G.log | cut -c1-400
