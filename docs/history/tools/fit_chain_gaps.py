"""Dependency chain of the TPS band reduction from a rocprofv3 kernel trace: gaps between the dependent launches
(panel -> symm -> first update block -> next panel), per-kernel durations, and the durations at a few trailing
sizes with the bandwidth they amount to.  Run by tools/fit_chain_gaps.sh on the GPU box."""
import collections, csv, glob

f = glob.glob('/tmp/pf/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))


def short(n):
    if 'band_update_kernel<true>' in n or 'band_update_kernelILb1' in n: return 'update_first'
    if 'band_update' in n: return 'update_rest'
    for k in ('band_panel_reg', 'band_symm', 'band_backtransform', 'band_extract', 'gram', 'symv', 'syr2', 'house_w'):
        if k in n: return k
    return n[:30]


start = [i for i, r in enumerate(rows) if 'gram_kernel' in r['Kernel_Name']][-1]      # the last fit of the run
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in rows[start:]]
print("kernels in the last fit:", len(ev), " span %.2f ms" % ((ev[-1][1] - ev[0][0]) / 1e6))
chain = [e for e in ev if e[2] in ('band_panel_reg', 'band_symm', 'update_first')]
gaps = collections.defaultdict(list)
for a, b in zip(chain[:-1], chain[1:]):
    gaps[a[2] + ' -> ' + b[2]].append(b[0] - a[1])
for k, v in gaps.items():
    v2 = sorted(v)
    print(f"{k:36s} n={len(v):4d} mean gap {sum(v) / len(v) / 1e3:7.2f} us  median {v2[len(v2) // 2] / 1e3:7.2f}")
durs = collections.defaultdict(list)
for s, e, n in ev:
    durs[n].append(e - s)
for k, v in durs.items():
    print(f"dur {k:22s} n={len(v):5d} mean {sum(v) / len(v) / 1e3:8.2f} us  total {sum(v) / 1e6:8.2f} ms")
pan = [e for e in ev if e[2] == 'band_panel_reg']
sym = [e for e in ev if e[2] == 'band_symm']
uf = [e for e in ev if e[2] == 'update_first']
ur = [e for e in ev if e[2] == 'update_rest']
m = 8 * len(pan) + 8 + 5      # unknowns of the fit (approximately)
print("panel: rows t, panel us, symm us (GB/s read), first update us, rest of the update us (GB/s read + write)")
for p in (0, 50, 100, 200, 300, 400, 500, 600):
    if p >= len(pan) or p >= len(ur): break
    t = m - 8 * p - 8
    sd, ud = (sym[p][1] - sym[p][0]) / 1e3, (ur[p][1] - ur[p][0]) / 1e3
    print(p, t, round((pan[p][1] - pan[p][0]) / 1e3, 1), round(sd, 1), round(t * t * 8 / sd / 1e3),
          round((uf[p][1] - uf[p][0]) / 1e3, 1), round(ud, 1), round(t * (t - 64) * 16 / ud / 1e3))
