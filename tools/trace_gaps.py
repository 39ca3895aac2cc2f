"""Per-queue view of ONE train step from a rocprofv3 kernel trace (tools/gpu_trace.sh writes gpurun_out/trace/t_kernel_trace.csv):
idle gaps of the main queue by (previous kernel -> next kernel), the timeline around the big ones, and per-phase kernel totals
(forward / backward on the main queue, weight-gradient queue).  Usage: python tools/trace_gaps.py [trace.csv]"""
import sys
TRACE = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/trace/t_kernel_trace.csv"
import csv, re, collections
rows=[r for r in csv.DictReader(open(TRACE))]
for r in rows: r['s']=int(r['Start_Timestamp']); r['e']=int(r['End_Timestamp'])
rows.sort(key=lambda r:r['s'])
idx=[i for i,r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
step=rows[idx[-2]+1:idx[-1]+1]
print(len(step),"kernels; wall", (step[-1]['e']-step[0]['s'])/1e3,"us")
def short(n):
    m=re.search(r'(\w+_kernel)',n); return m.group(1) if m else n[:40]
byq=collections.defaultdict(list)
for r in step: byq[r['Queue_Id']].append(r)
for q,rs in byq.items():
    busy=sum(r['e']-r['s'] for r in rs)/1e3
    print("queue",q,len(rs),"kernels busy %.1f us span %.1f"%(busy,(rs[-1]['e']-rs[0]['s'])/1e3))
# main queue = the one with most kernels
mq=max(byq,key=lambda q:len(byq[q])); rs=byq[mq]
gaps=[]
for a,b in zip(rs,rs[1:]):
    g=(b['s']-a['e'])/1e3
    gaps.append((g,short(a['Kernel_Name']),short(b['Kernel_Name'])))
tot=sum(g for g,_,_ in gaps if g>0)
print("main queue idle between kernels: %.1f us over %d gaps; negative overlaps %d"%(tot,len(gaps),sum(1 for g,_,_ in gaps if g<0)))
import statistics
pos=[g for g,_,_ in gaps if g>0]
print("median gap %.2f us, p90 %.2f, max %.2f"%(statistics.median(pos), sorted(pos)[int(.9*len(pos))], max(pos)))
agg=collections.defaultdict(lambda:[0,0.0])
for g,a,b in gaps:
    if g>0: agg[(a,b)][0]+=1; agg[(a,b)][1]+=g
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:25]:
    print("%8.1f us  n=%3d  avg %5.2f   %s -> %s"%(v[1],v[0],v[1]/v[0],k[0],k[1]))
# union busy of all queues
ev=sorted([(r['s'],1) for r in step]+[(r['e'],-1) for r in step])
act=0; last=None; busy=0
for t,d in ev:
    if act>0: busy+=t-last
    act+=d; last=t
print("any-queue busy %.1f us"%(busy/1e3))
print("---- timeline around big gaps")
t0=step[0]['s']
allr=sorted(step,key=lambda r:r['s'])
big=[i for i,(a,b) in enumerate(zip(rs,rs[1:])) if (b['s']-a['e'])>20000]
for i in big:
    a,b=rs[i],rs[i+1]
    lo,hi=a['s']-150000,b['e']+50000
    print("== gap %.1f us between %s and %s"%((b['s']-a['e'])/1e3,short(a['Kernel_Name']),short(b['Kernel_Name'])))
    for r in allr:
        if r['e']>=lo and r['s']<=hi:
            print("   q%s %9.1f -> %9.1f (%6.1f) %s grid %s"%(r['Queue_Id'],(r['s']-t0)/1e3,(r['e']-t0)/1e3,(r['e']-r['s'])/1e3,short(r['Kernel_Name'])[-28:],r['Grid_Size_X']))

print('---- per-phase kernel totals')
import csv, re, collections
rows=[r for r in csv.DictReader(open(TRACE))]
for r in rows: r['s']=int(r['Start_Timestamp']); r['e']=int(r['End_Timestamp'])
rows.sort(key=lambda r:r['s'])
idx=[i for i,r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
step=rows[idx[-2]+1:idx[-1]+1]
def short(n):
    n=n.replace('_ZN3seg12_GLOBAL__N_1','')
    m=re.match(r'\d+(\w+?)I(.*?)E+v', n)
    if m: return m.group(1)+'<'+m.group(2)+'>'
    m=re.search(r'(\w+_kernel)',n); return m.group(1) if m else n[:40]
t0=step[0]['s']
# find loss kernel time = boundary fwd/bwd
tl=[r for r in step if 'loss_reduce' in r['Kernel_Name']][0]['s']
for name,sel in (("FORWARD (main)",lambda r:r['s']<tl and r['Queue_Id']=='1'),("BACKWARD main",lambda r:r['s']>=tl and r['Queue_Id']=='1'),("SIDE",lambda r:r['Queue_Id']!='1')):
    rs=[r for r in step if sel(r)]
    if not rs: continue
    agg=collections.defaultdict(lambda:[0,0.0])
    for r in rs: a=agg[short(r['Kernel_Name'])]; a[0]+=1; a[1]+=(r['e']-r['s'])/1e3
    print("==",name,"%d kernels busy %.0f us span %.0f us"%(len(rs),sum(v[1] for v in agg.values()),(rs[-1]['e']-rs[0]['s'])/1e3))
    for k,v in sorted(agg.items(),key=lambda kv:-kv[1][1])[:14]: print("   %7.1f us n=%3d  %s"%(v[1],v[0],k[:70]))

print('---- full timeline of the step (queue, start us, duration us, idle before on the same queue, kernel, grid x*y*z / workgroup)')
lastq = {}
for r in sorted(step, key=lambda r: r['s']):
    q = r['Queue_Id']
    gap = (r['s'] - lastq[q]) / 1e3 if q in lastq else 0.0
    lastq[q] = r['e']
    wg = int(r.get('Workgroup_Size_X', 1) or 1)
    gx = int(r['Grid_Size_X']) // max(wg, 1)
    print("q%s %8.1f %7.1f %6.1f  %-58s %dx%sx%s/%d" % (q, (r['s'] - t0) / 1e3, (r['e'] - r['s']) / 1e3, gap, short(r['Kernel_Name'])[:58], gx,
                                                      r.get('Grid_Size_Y', '1'), r.get('Grid_Size_Z', '1'), wg))
