"""Level-generator kernels of a `rocprofv3 --kernel-trace --output-format csv -d DIR` run of bench.py, split by stream:
per step, the time of the kernels on the main stream (installs, and levels generated inside the step when an env's prefetch
slot was not refilled in time) and on the side stream (the prefetch generator).

    python tools/probe/level_split.py DIR
"""
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# main queue = the queue of render kernels
q_render=collections.Counter(r['Queue_Id'] for r in rows if 'render_kernel' in r['Kernel_Name']).most_common(1)[0][0]
tot=collections.defaultdict(float); cnt=collections.Counter(); big=0; bigt=0
for r in rows:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    n=r['Kernel_Name']
    key=('level' if 'level_kernel' in n else 'render' if 'render' in n else 'other')+(':main' if r['Queue_Id']==q_render else ':side')
    tot[key]+=d; cnt[key]+=1
    if key=='level:main' and d>50: big+=1; bigt+=d
steps=cnt['render:main']
print({k:(round(v/steps,1),cnt[k]) for k,v in tot.items()}, 'steps',steps)
print('main-stream level kernels > 50 us:',big,'of',cnt['level:main'],'adding',round(bigt/steps,1),'us per step')
