#!/bin/bash
# usage: [PG_GAME=coinrun] [PG_LIB=…] [PG_DEBUG=…] tools/probe/kernel_clocks.sh TAG
# What clock does each kernel of a step run at?  One rocprofv3 pass with --kernel-trace --pmc GRBM_GUI_ACTIVE over
# tools/pmc_quick.py: per kernel name, mean duration (ns, from the trace) and mean GRBM_GUI_ACTIVE (cycles the GPU was
# busy during the dispatch) over the last launches -> GHz.  A kernel behind a vector-heavy one may be handed a lower
# clock than the same kernel behind idle-ish latency chains (MI355X_MICROARCH.md: DVFS give-back).
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16
rm -rf /tmp/kc_$$
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/kc_$$ -- python3 $R/tools/pmc_quick.py > /tmp/kc_$$.log 2>&1
python3 - /tmp/kc_$$ "$TAG" <<'PY'
import csv, glob, sys, collections
d, tag = sys.argv[1], sys.argv[2]
dur = {}
for fn in glob.glob(d + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(fn)):
        dur[int(r["Dispatch_Id"])] = (r["Kernel_Name"].split("(")[0], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
cyc = {}
for fn in glob.glob(d + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cyc[int(r["Dispatch_Id"])] = float(r["Counter_Value"])
per = collections.defaultdict(list)
for k in sorted(dur):
    if k in cyc: per[dur[k][0]].append((dur[k][1], cyc[k]))
print("== %s" % tag)
for name, v in per.items():
    v = v[-8:]
    ns = sum(a for a, _ in v) / len(v); c = sum(b for _, b in v) / len(v)
    print("%-60s %9.1f us  %12.0f cycles  %.3f GHz" % (name[-60:], ns / 1e3, c, c / ns if ns else 0))
PY
