#!/bin/bash
# usage: [PG_GAME=coinrun] [PG_LIB=procgen2_amd/lib/x.so] tools/pmc_quick.sh TAG "SET1 COUNTERS" "SET2 COUNTERS" …
# one rocprofv3 --pmc pass per quoted set (no trace domain besides --kernel-trace); prints the render kernel's
# per-launch averages over the last 8 launches and writes gpurun_out/TAG_pmc.json
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
mkdir -p $(dirname $R/gpurun_out/${TAG}_x)
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16   # the profiler initialises HIP before the library can: same configuration as the bench line
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
result = collections.OrderedDict()
for kernel in "${PG_KERNEL:-render_kernel}".split(","):   # (several kernels: comma-separated)
    out = collections.OrderedDict()
    for d in sorted(glob.glob("/tmp/pq_[0-9]*/")):
        rows = collections.defaultdict(dict)
        for fn in glob.glob(d + "*/*counter_collection.csv"):
            for r in csv.DictReader(open(fn)):
                if kernel not in r["Kernel_Name"]: continue
                rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
        order = sorted(rows)[-8:]
        for c in sorted({k for d_ in order for k in rows[d_]}):
            out[c] = sum(rows[d_].get(c, 0.0) for d_ in order) / max(1, len(order))
    print("== " + kernel)
    for k, v in out.items(): print("%-36s %.6g" % (k, v))
    if "SQ_WAVES" in out:
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"):
            if k in out: print("  %-34s %.1f per wave" % (k, out[k] / out["SQ_WAVES"]))
    result[kernel] = out
json.dump(result if len(result) > 1 else list(result.values())[0], open("$R/gpurun_out/${TAG}_pmc.json", "w"), indent=1)
PY
