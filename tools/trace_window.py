#!/usr/bin/env python3
"""Timeline around the k-th last evaluation kernel (= the boundary between two pipelined proofs) of a rocprofv3 --kernel-trace run.
Usage: python tools/trace_window.py <dir with *kernel_trace.csv> [k=3] [before_ms=45] [after_ms=30] [min_ms=0.2]"""
import csv, glob, os, re, sys
files = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True)
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
before, after = (float(sys.argv[3]) if len(sys.argv) > 3 else 45.0), (float(sys.argv[4]) if len(sys.argv) > 4 else 30.0)
min_ms = float(sys.argv[5]) if len(sys.argv) > 5 else 0.2
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')))
rows.sort()
def short(n):
    n = re.sub(r'\(.*$', '', n).replace('void ', '').replace('fk::', '')
    n = n.replace('Fp<FqParams, true>', 'Fq').replace('Fq2T<Fq >', 'Fq2').replace('Fp<FqParams, false>', 'FqC').replace('Fp<FrParams, true>', 'Fr').replace('Fp<FrLazyParams, true>', 'FrL')
    return n[:48]
sp = [r for r in rows if 'spmv_binned' in r[2]]
t0 = sp[-k][0]
sel = [r for r in rows if r[1] >= t0 - before * 1e6 and r[0] <= t0 + after * 1e6]
merged = []
for s, e, n, q in sel:
    n = short(n)
    if merged and merged[-1][2] == n and merged[-1][3] == q and s - merged[-1][1] < 200000:
        merged[-1][1] = e; merged[-1][4] += 1
    else:
        merged.append([s, e, n, q, 1])
qs = sorted(set(m[3] for m in merged))
print('t = 0: start of the evaluation kernel of the %d-th last proof; queues' % k, qs)
for s, e, n, q, c in merged:
    d = (e - s) / 1e6
    if d < min_ms: continue
    print('%8.2f %8.2f  q%-3s %-50s x%d' % ((s - t0) / 1e6, d, qs.index(q), n, c))
