#!/bin/bash
# usage: [PG_GAME=coinrun] [PG_LIB=procgen2_amd/lib/x.so] tools/pmc_quick.sh TAG "SET1 COUNTERS" "SET2 COUNTERS" …
# one rocprofv3 --pmc pass per quoted set (no trace domain besides --kernel-trace); prints the render kernel's
# per-launch averages over the last 8 launches and writes gpurun_out/TAG_pmc.json
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
mkdir -p $(dirname $R/gpurun_out/${TAG}_x)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pq_[0-9]*
i=0
for SET in "$@"; do
  rm -rf /tmp/pq_$i
  timeout ${PG_PASS_TIMEOUT:-120} rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pq_$i -- python3 $R/tools/pmc_quick.py > /tmp/pq_$i.log 2>&1
  echo "pass $i ($SET): rc=$?" >&2
  i=$((i+1))
done
python3 - <<PY
import csv, glob, json, collections
out = collections.OrderedDict()
for d in sorted(glob.glob("/tmp/pq_[0-9]*/")):
    rows = collections.defaultdict(dict)
    for fn in glob.glob(d + "*/*counter_collection.csv"):
        for r in csv.DictReader(open(fn)):
            if "${PG_KERNEL:-render_kernel}" not in r["Kernel_Name"]: continue
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    order = sorted(rows)[-8:]
    for c in sorted({k for d_ in order for k in rows[d_]}):
        out[c] = sum(rows[d_].get(c, 0.0) for d_ in order) / max(1, len(order))
for k, v in out.items(): print("%-36s %.6g" % (k, v))
json.dump(out, open("$R/gpurun_out/${TAG}_pmc.json", "w"), indent=1)
PY
