#!/bin/bash
# usage: tools/ablate_counters.sh TAG  → gpurun_out/TAG_ablate_counters.json  (needs lib/libprocgen2_hip_ablate.so)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}
mkdir -p $(dirname $R/gpurun_out/${TAG}_x)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/abl_$TAG
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d /tmp/abl_$TAG -- python3 $R/tools/ablate_counters.py run > $R/gpurun_out/${TAG}_ablate.log 2>&1
python3 - <<PY
import csv, glob, json, collections
plan = json.loads([l for l in open("$R/gpurun_out/${TAG}_ablate.log") if l.startswith('{"settle"')][-1])
rows = collections.defaultdict(dict)
for fn in glob.glob("/tmp/abl_$TAG/*/*counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        if "render_kernel" not in r["Kernel_Name"]: continue
        rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
order = sorted(rows)
at = plan["settle"]
out = []
for g in plan["groups"]:
    sel = order[at:at + g["launches"]]; at += g["launches"]
    agg = collections.defaultdict(float)
    for d in sel:
        for k, v in rows[d].items(): agg[k] += v / len(sel)
    w = agg.get("SQ_WAVES", 1.0) or 1.0
    g.update({k: v for k, v in agg.items()})
    g.update({"valu_per_wave": agg["SQ_INSTS_VALU"] / w, "salu_per_wave": agg["SQ_INSTS_SALU"] / w,
              "lds_per_wave": agg["SQ_INSTS_LDS"] / w, "vmem_rd_per_wave": agg["SQ_INSTS_VMEM_RD"] / w})
    out.append(g)
    print("%-24s %.3f ms  VALU/wave %7.1f  SALU/wave %6.1f  LDS/wave %5.1f  VMEM_RD/wave %5.1f" % (g["name"], g["render_ms"], g["valu_per_wave"], g["salu_per_wave"], g["lds_per_wave"], g["vmem_rd_per_wave"]))
json.dump({"render_launches_total": len(order), "groups": out}, open("$R/gpurun_out/${TAG}_ablate_counters.json", "w"), indent=1)
PY
