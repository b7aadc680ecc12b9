#!/bin/bash
# usage: tools/bound_table.sh TAG [game …]
# Counter passes for the bound table: every kernel of every game's step, 65 536 envs, steady state
# (tools/pmc_quick.py: 300 settle steps, then 8 measured), one rocprofv3 --kernel-trace --pmc pass per counter set (the TCC
# sets alone, as MI355X_MICROARCH.md asks).  Output: gpurun_out/TAG_bounds_raw.json — per game and kernel, per-launch
# averages over the last 8 launches + the mean duration from the trace.  tools/bound_table.py turns it into the table.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
GAMES=${@:-coinrun maze bossfight climber caveflyer chaser jumper}
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16
SETS=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES"
 "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"
 "TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum"
 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
 "TD_TD_BUSY_sum TD_TC_STALL_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"
)
for G in $GAMES; do
  i=0
  for SET in "${SETS[@]}"; do
    rm -rf /tmp/bt_${G}_$i
    PG_GAME=$G timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/bt_${G}_$i -- python3 $R/tools/pmc_quick.py > /tmp/bt_${G}_$i.log 2>&1 || echo "pass $i of $G failed ($SET)" >&2
    i=$((i+1))
  done
done
python3 - "$R/gpurun_out/${TAG}_bounds_raw.json" $GAMES <<'PY'
import csv, glob, json, sys, collections
out_path, games = sys.argv[1], sys.argv[2:]
result = {}
for g in games:
    per_kernel = collections.defaultdict(dict)
    for d in sorted(glob.glob("/tmp/bt_%s_[0-9]*/" % g)):
        rows = collections.defaultdict(lambda: collections.defaultdict(dict))  # kernel -> dispatch -> counter
        dur = collections.defaultdict(dict)
        for fn in glob.glob(d + "*/*counter_collection.csv"):
            for r in csv.DictReader(open(fn)):
                k = r["Kernel_Name"].split("(")[0]
                rows[k][int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
        for fn in glob.glob(d + "*/*kernel_trace.csv"):
            for r in csv.DictReader(open(fn)):
                dur[r["Kernel_Name"].split("(")[0]][int(r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for k, by_dispatch in rows.items():
            if "::%s::" % g not in k: continue
            last = sorted(by_dispatch)[-8:]
            for c in {c for x in last for c in by_dispatch[x]}:
                per_kernel[k][c] = sum(by_dispatch[x].get(c, 0.0) for x in last) / len(last)
            ns = [dur[k][x] for x in last if x in dur[k]]
            if ns: per_kernel[k].setdefault("_ns", []).append(sum(ns) / len(ns))
    for k in per_kernel:
        v = per_kernel[k].pop("_ns", [])
        per_kernel[k]["duration_ns"] = sum(v) / len(v) if v else 0.0
    result[g] = per_kernel
json.dump(result, open(out_path, "w"), indent=1)
print("wrote", out_path, {g: len(result[g]) for g in result})
PY
