#!/bin/bash
# HBM traffic of the step kernels from PMC counters: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots:
# MI355X_MICROARCH.md §rocprofv3 PMC slots), each with --kernel-trace only.  Output: gpurun_out/<tag>_traffic.json
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/${TAG//\//_}_pmc_$C -- python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline > $R/gpurun_out/${TAG}_pmc_$C.log 2>&1
done
python3 - <<PY
import csv,glob,collections,json
out={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    agg=collections.defaultdict(float); cnt=collections.Counter()
    for fn in glob.glob("/tmp/${TAG//\//_}_pmc_%s/*/*counter_collection.csv"%c):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"]!=c: continue
            k=r["Kernel_Name"].split("(")[0]
            agg[k]+=float(r["Counter_Value"]); cnt[k]+=1
    for k in agg:
        if "coinrun" in k: out.setdefault(k,{})[c+"_KB_per_launch"]=agg[k]/cnt[k]
for k,v in out.items():
    f=v.get("FETCH_SIZE_KB_per_launch",0.0); w=v.get("WRITE_SIZE_KB_per_launch",0.0)
    v["bytes_raw"]=(f+w)*1024
    v["bytes_corrected"]=(2*f+w)*1024   # gfx950: FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md §HBM)
import sys
sys.path.insert(0,"$R")
import bench
out["_csrc_sha1"]=bench.csrc_fingerprint()   # what the profile was taken on: bench.py reports traffic_stale when the tree has moved on
json.dump(out,open("$R/gpurun_out/${TAG}_traffic.json","w"),indent=1)
print(json.dumps(out,indent=1))
PY
