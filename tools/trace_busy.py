#!/usr/bin/env python3
"""GPU busy fraction of a rocprofv3 --kernel-trace run: union of the kernel intervals over the span they cover, and the
per-proof gap structure.  Usage: python tools/trace_busy.py <dir with *kernel_trace.csv>"""
import csv
import glob
import os
import sys

files = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True)
if not files:
    raise SystemExit('no kernel_trace.csv under ' + sys.argv[1])
iv = []
for f in files:
    for r in csv.DictReader(open(f)):
        iv.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
iv.sort()
# the timed region = the last 60 % of the span (warm-up, setup and level precomputation come first)
t0, t1 = iv[0][0], max(e for _, e, _ in iv)
lo = t0 + (t1 - t0) * 0.4
sel = [(s, e) for s, e, _ in iv if s >= lo]
busy, cur_s, cur_e, gaps = 0, None, None, []
for s, e in sel:
    if cur_e is None:
        cur_s, cur_e = s, e
    elif s <= cur_e:
        cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s; gaps.append(s - cur_e); cur_s, cur_e = s, e
busy += cur_e - cur_s
span = max(e for _, e in sel) - sel[0][0]
gaps.sort(reverse=True)
print('kernels in the window: %d, span %.1f ms, GPU busy (union of kernel intervals) %.1f ms = %.1f %%' % (len(sel), span / 1e6, busy / 1e6, 100.0 * busy / span))
print('idle gaps: %d, total %.1f ms; largest (ms): %s' % (len(gaps), sum(gaps) / 1e6, ', '.join('%.2f' % (g / 1e6) for g in gaps[:12])))
