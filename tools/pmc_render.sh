#!/bin/bash
# PMC pass over a short bench run (counters only: no trace domains beside --kernel-trace).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16   # the profiler initialises HIP before the library can: same configuration as the bench line
timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -- python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline ${PG_BENCH_ARGS} > $OUT.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/*/*counter_collection.csv")
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k in agg:
    if "render" in k or "logic" in k or "entity" in k or "agent_kernel" in k:
        print(k)
        for c,v in sorted(agg[k].items()): print("   %-24s %.4g per launch"%(c, v/cnt[(k,c)]))
PY
