#!/bin/bash
# HBM traffic of every game's step kernels from PMC counters: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots:
# MI355X_MICROARCH.md §rocprofv3 PMC slots), each with --kernel-trace only, over a short bench.py run of the game at
# 65 536 envs in steady state.  Output: gpurun_out/<tag>_traffic.json — per kernel and launch, with the gfx950
# FETCH_SIZE correction and the fingerprint of the kernel sources; bench.py reads `roofline.traffic` of ANY --game from
# the newest profiles/*traffic_pmc.json (copy it there).
# usage: tools/pmc_traffic.sh TAG [game …]     (default: all seven)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}; shift
GAMES=${@:-coinrun maze bossfight climber caveflyer chaser jumper}
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16   # the profiler initialises HIP before the library can: same configuration as the bench line
for G in $GAMES; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/${TAG//\//_}_pmc_${G}_$C
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/${TAG//\//_}_pmc_${G}_$C -- python3 $R/bench.py --game $G --steps 16 --warmup 4 --no-cpu-baseline > $R/gpurun_out/${TAG}_pmc_${G}_$C.log 2>&1
  done
done
python3 - <<PY
import csv,glob,collections,json
out={}
for g in "$GAMES".split():
    for c in ("FETCH_SIZE","WRITE_SIZE"):
        agg=collections.defaultdict(float); cnt=collections.Counter()
        for fn in glob.glob("/tmp/${TAG//\//_}_pmc_%s_%s/*/*counter_collection.csv"%(g,c)):
            for r in csv.DictReader(open(fn)):
                if r["Counter_Name"]!=c: continue
                k=r["Kernel_Name"].split("(")[0]
                agg[k]+=float(r["Counter_Value"]); cnt[k]+=1
        for k in agg:
            if "::%s::"%g in k: out.setdefault(k,{})[c+"_KB_per_launch"]=agg[k]/cnt[k]
for k,v in out.items():
    f=v.get("FETCH_SIZE_KB_per_launch",0.0); w=v.get("WRITE_SIZE_KB_per_launch",0.0)
    v["bytes_raw"]=(f+w)*1024
    v["bytes_corrected"]=(2*f+w)*1024   # gfx950: FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md §HBM)
import sys
sys.path.insert(0,"$R")
import bench
out["_csrc_sha1"]=bench.csrc_fingerprint()   # what the profile was taken on: bench.py reports traffic_stale when the tree has moved on
out["_note"]="rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) of bench.py --game G --steps 16 --warmup 4 after its 512 settle steps; averages per launch over all launches of the run"
json.dump(out,open("$R/gpurun_out/${TAG}_traffic.json","w"),indent=1)
for k,v in out.items():
    if isinstance(v,dict) and ("render_kernel" in k or "setup_kernel" in k): print(k, "%.0f MB corrected (%.2f x algorithmic)"%(v["bytes_corrected"]/1e6, v["bytes_corrected"]/(65536*12297)))
PY
