import csv,glob,sys
f=glob.glob('/tmp/kt/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=rows[-24:]
t0=int(rows[0]['Start_Timestamp'])
for r in rows:
    n=r['Kernel_Name']
    n='render' if 'render' in n else 'logic' if 'logic' in n else 'level' if 'level' in n else n[:20]
    print('%-8s start %8.1f us  dur %7.1f us  stream/queue %s' % (n,(int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r.get('Queue_Id','')))
