"""The last kernels of a `rocprofv3 --kernel-trace --output-format csv -d DIR` run as a timeline (start, duration, queue).

    python tools/trace_timeline.py DIR [N]        # DIR as given to rocprofv3 -d; N = how many kernels (default 24)
"""
import csv
import glob
import os
import sys

if len(sys.argv) < 2:
    sys.exit(__doc__)
files = glob.glob(os.path.join(sys.argv[1], "*", "*kernel_trace.csv")) + glob.glob(os.path.join(sys.argv[1], "*kernel_trace.csv"))
if not files:
    sys.exit("no *kernel_trace.csv under %s" % sys.argv[1])
count = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-count:]
t0 = int(rows[0]['Start_Timestamp'])
for r in rows:
    n = r['Kernel_Name']
    n = 'render' if 'render' in n else 'setup' if 'setup' in n else 'logic' if 'logic' in n else 'level' if 'level' in n else n[:20]
    print('%-8s start %8.1f us  dur %7.1f us  stream/queue %s' % (n, (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Queue_Id', '')))
