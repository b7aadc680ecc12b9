#!/usr/bin/env python3
"""Summary of a rocprofv3 --kernel-trace of `bench.py --workload mixed`: for the last steps of the run, per kernel its
stream, start and duration, and per step how busy the device was (union of kernel intervals ÷ step time) and how many
kernels ran at once on average.  Usage: python tools/mixed_timeline.py <kernel_trace.csv> [out.json]"""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    game = n.split("::")[2] if n.startswith("pg::variant") else ("level:" + n.split("variant0::")[1].split("::")[0] if "level_kernel" in n else n[:16])
    kind = "render" if "render_kernel" in n else "logic" if "logic_kernel" in n else "level" if "level_kernel" in n else n.split("::")[-1].split("(")[0]
    return game + "/" + kind


tail = rows[-2200:]  # ≈ 60 steps of 36 kernels
t0 = int(tail[0]["Start_Timestamp"])
ivals = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, short(r["Kernel_Name"]), r.get("Queue_Id", "")) for r in tail]
span = ivals[-1][1]
# union length and overlap depth
events = sorted([(a, 1) for a, b, _, _ in ivals] + [(b, -1) for a, b, _, _ in ivals])
busy = depth_time = 0
depth = 0
last = 0
for t, d in events:
    if depth > 0:
        busy += t - last
        depth_time += depth * (t - last)
    depth += d
    last = t
per_kernel = {}
for a, b, n, q in ivals:
    k = per_kernel.setdefault(n, {"calls": 0, "total_us": 0.0, "queue": q})
    k["calls"] += 1
    k["total_us"] += (b - a) / 1e3
for k in per_kernel.values():
    k["avg_us"] = k["total_us"] / k["calls"]
out = {"window_us": span / 1e3, "device_busy_fraction": busy / span, "mean_kernels_in_flight_while_busy": depth_time / max(1, busy),
       "sum_of_kernel_time_over_window": sum(b - a for a, b, _, _ in ivals) / span,
       "kernels": dict(sorted(per_kernel.items(), key=lambda kv: -kv[1]["total_us"]))}
print(json.dumps(out, indent=1)[:6000])
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
