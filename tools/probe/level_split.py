"""Level-generator kernels of a `rocprofv3 --kernel-trace --output-format csv -d DIR` run of bench.py, split by stream:
per step, the time of the kernels on the main stream (installs, and levels generated inside the step when an env's prefetch
slot was not refilled in time) and on the side stream (the prefetch generator).  The first 64 main-stream level kernels
(cenv_make, the first reset and the steps right after them, when no slot has been filled yet) are listed apart: they are
the start-up, not the steady state.

    python tools/probe/level_split.py DIR
"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
q_render = collections.Counter(r['Queue_Id'] for r in rows if 'render_kernel' in r['Kernel_Name']).most_common(1)[0][0]
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
main = [dur(r) for r in rows if 'level_kernel' in r['Kernel_Name'] and r['Queue_Id'] == q_render]
side = [dur(r) for r in rows if 'level_kernel' in r['Kernel_Name'] and r['Queue_Id'] != q_render]
steps = len(main)
start, steady = main[:64], main[64:]
long_steady = [d for d in steady if d > 50]
print('%d steps; side-stream generator: %d launches, %.1f us per step' % (steps, len(side), sum(side) / steps))
print('start-up (first 64 main-stream level kernels): %.1f ms in all, longest %.1f ms' % (sum(start) / 1e3, max(start) / 1e3))
print('steady state: %.1f us per step on the main stream; of it %.1f us per step in %d kernels longer than 50 us (mean %.0f us, longest %.0f us)'
      % (sum(steady) / len(steady), sum(long_steady) / len(steady), len(long_steady),
         sum(long_steady) / max(1, len(long_steady)), max(long_steady) if long_steady else 0))
