#!/usr/bin/env python3
"""Timeline of the LAST proof in a rocprofv3 --kernel-trace run: one line per kernel launch (start and duration in ms relative
to the proof's first kernel, queue, short name), consecutive launches of the same kernel on the same queue merged.
Usage: python tools/trace_gantt.py <dir with *kernel_trace.csv> [min_ms]"""
import csv, glob, os, re, sys
files = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True)
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?'), r.get('Stream_Id', '?')))
rows.sort()
def short(n):
    n = re.sub(r'\(.*$', '', n).replace('void ', '').replace('fk::', '')
    n = n.replace('Fp<FqParams, true>', 'Fq').replace('Fq2T<Fq >', 'Fq2').replace('Fp<FqParams, false>', 'FqC').replace('Fp<FrParams, true>', 'Fr').replace('Fp<FrLazyParams, true>', 'FrL')
    return n[:48]
# proofs start with the SpMV kernel
# the last proof: from the last launch of the evaluation's first kernel (spmv_kernel: the input rows and the short matrices; the wave
# and length-class kernels follow it) to the end
i0 = [i for i, r in enumerate(rows) if 'spmv_kernel' in r[2]][-1]
sel = rows[i0:]
t0 = sel[0][0]
merged = []
for s, e, n, q, st in sel:
    n = short(n)
    if merged and merged[-1][2] == n and merged[-1][3] == q and s - merged[-1][1] < 200000:
        merged[-1][1] = e; merged[-1][4] += 1
    else:
        merged.append([s, e, n, q, 1])
qs = sorted(set(m[3] for m in merged))
print('queues:', qs, ' proof span %.2f ms' % ((max(m[1] for m in merged) - t0) / 1e6))
for s, e, n, q, k in merged:
    d = (e - s) / 1e6
    if d < min_ms: continue
    print('%8.2f %8.2f  q%-3s %-50s x%d' % ((s - t0) / 1e6, d, qs.index(q), n, k))
