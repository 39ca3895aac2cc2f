import sqlite3, re, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]; ci = {c:i for i,c in enumerate(cols)}
rows = list(cur.execute("select * from kernels order by start"))
names = [r[ci['name']] for r in rows]
idx = [i for i,n in enumerate(names) if 'adam_kernel' in n]
step = rows[idx[-2]+1:idx[-1]+1]
def short(n):
    n = n.replace('_ZN3seg12_GLOBAL__N_1','')
    m = re.match(r'\d+(\w+?)I(.*?)E+v', n)
    return (m.group(1)+'<'+m.group(2)+'>') if m else n[:50]
agg = {}
tot = 0
for r in step:
    n = short(r[ci['name']]); d = (r[ci['end']]-r[ci['start']])/1000.0; tot += d
    a = agg.setdefault(n, [0, 0.0, 0.0]); a[0]+=1; a[1]+=d; a[2]=max(a[2], d)
print("step GPU-busy total %.1f us, %d kernels; wall %.1f us" % (tot, len(step), (step[-1][ci['end']]-step[0][ci['start']])/1000.0))
for n,(c,d,mx) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:int(sys.argv[2]) if len(sys.argv)>2 else 22]:
    print("%9.1f us %5.1f%%  calls %3d  max %8.1f  %s" % (d, 100*d/tot, c, mx, n))
