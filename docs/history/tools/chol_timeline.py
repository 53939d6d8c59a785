"""Timeline of the blocked Cholesky from a rocprofv3 kernel trace (tools/r02_chol_timeline.sh): how much of the wall
time has an update kernel running, how much only chain kernels (diagonal block, panel solve), how much nothing."""
import csv, glob, sys
f = glob.glob('/tmp/ct/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
start = [i for i, r in enumerate(rows) if 'gram_' in r['Kernel_Name']][-1]      # the last fit's Gram / projection launch (gram_kernel or the fused gram_proj_kernel)
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].split('::')[-1]) for r in rows[start:]]
t0, t1 = ev[0][0], max(e[1] for e in ev)
print("kernels", len(ev), "span %.2f ms" % ((t1 - t0) / 1e6))
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = None, None
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: tot += ce - cs; cs, ce = s, e
    if cs is not None: tot += ce - cs
    return tot
names = sorted({e[2] for e in ev})
for n in names:
    iv = [(s, e) for s, e, k in ev if k == n]
    print("%-28s n=%5d sum %8.2f ms  union %8.2f ms" % (n, len(iv), sum(e - s for s, e in iv) / 1e6, union(iv) / 1e6))
syrk = [(s, e) for s, e, k in ev if 'syrk' in k]
allk = [(s, e) for s, e, k in ev]
print("any kernel running: %.2f ms; syrk running: %.2f ms; idle: %.2f ms" % (union(allk) / 1e6, union(syrk) / 1e6, (t1 - t0 - union(allk)) / 1e6))
chain = [(s, e) for s, e, k in ev if 'diag' in k or 'trsm' in k]
# time where chain kernels run but no syrk
import bisect
def subtract(a, b):   # total length of a not covered by b (both lists of intervals)
    b = sorted(b); merged = []
    for s, e in b:
        if merged and s <= merged[-1][1]: merged[-1][1] = max(merged[-1][1], e)
        else: merged.append([s, e])
    tot = 0
    for s, e in sorted(a):
        cur = s
        for ms, me in merged:
            if me <= cur: continue
            if ms >= e: break
            if ms > cur: tot += ms - cur
            cur = max(cur, me)
            if cur >= e: break
        if cur < e: tot += e - cur
    return tot
print("chain kernels with no syrk beside them: %.2f ms" % (subtract(chain, syrk) / 1e6))
