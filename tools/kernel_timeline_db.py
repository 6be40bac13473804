import sqlite3, re, sys, collections
db=sqlite3.connect(sys.argv[1])
rows=db.execute("select name, start, end from kernels order by start").fetchall()
def short(n):
    m=re.search(r'(k_\w+)', n); return m.group(1) if m else n[:30]
prim=[i for i,r in enumerate(rows) if 'k_primary' in r[0]]
i0,i1=prim[-3],prim[-2]
t0=rows[i0][1]
for r in rows[i0:i1]:
    print(f"{short(r[0]):12s} start {(r[1]-t0)/1e3:7.1f} dur {(r[2]-r[1])/1e3:7.1f}")
print("period", (rows[i1][1]-t0)/1e3)
