#!/usr/bin/env python3
"""Per kernel name: launches, SUM of the launch durations (what rocprofv3 --stats reports) and UNION of the launch intervals (the
time during which the kernel was running at all) over a rocprofv3 --kernel-trace run.  Launches of one kernel that run side by
side on different streams -- the G1 accumulations of B1, L and A in the sorts-first schedule -- count once in the union.
Usage: python tools/trace_union.py <dir with *kernel_trace.csv> [proofs in the run | auto]
auto: the number of proofs is derived from the trace itself -- launches of the G1 bucket accumulation / 4 (H, L, A, B1)."""
import collections, csv, glob, os, re, sys
files = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True)
proofs = float(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != 'auto' else 1.0
iv = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        n = re.sub(r'\(.*$', '', r['Kernel_Name']).replace('void ', '').replace('fk::', '')
        n = n.replace('Fp<FqParams, true>', 'Fq').replace('Fq2T<Fq >', 'Fq2').replace('Fp<FqParams, false>', 'FqC').replace('Fp<FrLazyParams, true>', 'FrL')
        iv[n].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
if len(sys.argv) > 2 and sys.argv[2] == 'auto':
    acc = sum(len(v) for n, v in iv.items() if n.startswith(('msm_accumulate_merged_kernel<Fq,', 'msm_accumulate_kernel<Fq,')))
    assert acc and acc % 4 == 0, 'G1 accumulate launches: %d' % acc
    proofs = acc / 4.0
rows = []
for n, v in iv.items():
    v.sort()
    tot = sum(e - s for s, e in v)
    u, cs, ce = 0, None, None
    for s, e in v:
        if ce is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: u += ce - cs; cs, ce = s, e
    u += ce - cs
    rows.append((u, tot, len(v), n))
rows.sort(reverse=True)
print('%-58s %8s %14s %14s   (ms per proof, %g proofs)' % ('kernel', 'launches', 'sum', 'union', proofs))
for u, tot, k, n in rows[:28]:
    print('%-58s %8d %14.2f %14.2f' % (n[:58], k, tot / 1e6 / proofs, u / 1e6 / proofs))
