#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc counter_collection CSVs (one directory per pass) into the JSON files under profiles/.

    python tools/pmc_summary.py traffic  <fetch_dir> <write_dir> <log2n> <points_per_proof> <out.json> [proofs in the pass | auto] [workload] [gathercal dir] [rows]

The number of proofs in a pass is DERIVED from the pass itself: dispatches of the G1 accumulation / 4 (H, L, A, B1 -- one launch
each per proof), cross-checked against the dispatches of ntt_pass_kernel (6 transforms x ceil(log2n / 9) passes per proof, plus the one or two
transforms of the pass's key set-up) in both the FETCH and the WRITE pass.  A number given on the command line must agree with it (round 3 summarised a 5-proof pass with
the default of 1 and put 5x the traffic into bench.py's line).

    python tools/pmc_summary.py valu     <sq_dir> <derived_dir> <out.json>

FETCH_SIZE / WRITE_SIZE are in KB (x1024 = bytes).  FETCH_SIZE of wide coalesced streams under-reports by 2x on gfx950
(/opt/skills/guides/MI355X_MICROARCH.md); the gather width of the MSM accumulate kernel is uncalibrated, so the raw
figure is recorded and labelled as such.
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r'\(.*$', '', name)
    return name.replace('void ', '').replace('fk::', '')


def load(d):
    rows = []
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        rows += list(csv.DictReader(open(f)))
    if not rows:
        raise SystemExit('no counter_collection.csv under ' + d)
    return rows


def per_kernel(rows):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in rows:
        k = short(r['Kernel_Name'])
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        disp[k].add(r['Dispatch_Id'])
    return agg, {k: len(v) for k, v in disp.items()}


def gather_correction(gcal_dir):
    """demand / FETCH_SIZE of tools/mulbench/gathercal's 64-byte gather kernel (None without the pass)"""
    if not gcal_dir:
        return None, {}
    agg, _ = per_kernel(load(gcal_dir))
    blocks, threads, per = 256 * 16, 256, 64
    demand = {'gather_kernel<4>': blocks * threads * per * 64, 'gather_kernel<8>': blocks * threads * per * 128, 'stream_kernel': 4 << 30}
    res = {k: demand[k] / (agg[k]['FETCH_SIZE'] * 1024) for k in demand if k in agg and agg[k]['FETCH_SIZE'] > 0}
    return res.get('gather_kernel<4>'), res


def proofs_in_pass(launches, dom, log2n, what):
    """proofs in a pass, from its own dispatch counts"""
    n_acc = launches[dom]
    if n_acc % 4:
        raise SystemExit('%s pass: %d G1-accumulate dispatches is not a multiple of 4 (H, L, A, B1 per proof)' % (what, n_acc))
    proofs = n_acc // 4
    ntt = sum(v for k, v in launches.items() if k.startswith('ntt_pass_kernel'))
    ppt = (log2n + 8) // 9                       # passes per transform
    # 6 transforms per proof; the key set-up of the pass (fk_setup*: the Lagrange values are one inverse transform) adds one or two
    extra = ntt - proofs * 6 * ppt
    if ntt and (ntt % ppt or extra < 0 or extra > 2 * ppt):
        raise SystemExit('%s pass: %d ntt_pass_kernel dispatches do not fit %d proofs x 6 transforms x %d passes (+ at most 2 set-up transforms)'
                         % (what, ntt, proofs, ppt))
    return proofs


def traffic(fetch_dir, write_dir, log2n, points, out, proofs=None, workload='synthetic', gcal_dir=None, rows=None):
    fa, fl = per_kernel(load(fetch_dir))
    wa, wl = per_kernel(load(write_dir))
    ks = sorted(fa, key=lambda k: -(fa[k]['FETCH_SIZE'] + wa.get(k, {}).get('WRITE_SIZE', 0)))
    per = [dict(kernel=k, launches=fl[k], FETCH_SIZE_KB=fa[k]['FETCH_SIZE'], WRITE_SIZE_KB=wa.get(k, {}).get('WRITE_SIZE', 0.0)) for k in ks[:24]]
    # the G1 accumulation: the merged form (fixed-base levels, one bucket set) when the key carries levels, else the W-set form
    dom = next(k for k in ks if k.startswith(('msm_accumulate_merged_kernel<Fp<FqParams', 'msm_accumulate_kernel<Fp<FqParams')))
    merged = dom.startswith('msm_accumulate_merged')
    derived = proofs_in_pass(fl, dom, log2n, 'FETCH_SIZE')
    if proofs_in_pass(wl, dom, log2n, 'WRITE_SIZE') != derived:
        raise SystemExit('the FETCH_SIZE and WRITE_SIZE passes hold different numbers of proofs')
    if proofs is not None and proofs != derived:
        raise SystemExit('proofs given (%d) != proofs derived from the dispatches (%d)' % (proofs, derived))
    proofs = derived
    corr, corr_all = gather_correction(gcal_dir)
    j = dict(
        workload=workload, workload_is=('bench.py --workload %s%s: the pass is matched to a bench line by this name, the domain AND the row count' % (workload, (' (%d rows)' % int(rows)) if rows else '')), fetch_size_calibration=corr_all,
        _doc='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python3 bench.py --steps 1 --warmup 0 '
             '--no-cpu-baseline` (2^%d rows).  Counter units: KB (x1024 = bytes), summed over the launches of the whole pass (per_kernel); '
             'dominant_kernel is per proof.' % log2n,
        log2n=log2n, rows=rows, proofs_in_the_pass=proofs,
        proofs_in_the_pass_is='derived: G1-accumulate dispatches / 4, cross-checked against ntt_pass_kernel dispatches (6 transforms x ceil(log2n / 9) passes per proof)',
        per_kernel=per,
        dominant_kernel=dict(
            name='msm_accumulate_merged_kernel<Fq>' if merged else 'msm_accumulate_kernel<Fq>', launches_per_proof=fl[dom] // proofs, points_per_proof=points, proofs_in_the_pass=proofs,
            fetch_bytes_per_proof_raw=fa[dom]['FETCH_SIZE'] * 1024 / proofs, write_bytes_per_proof=wa[dom]['WRITE_SIZE'] * 1024 / proofs,
            fetch_correction=corr if corr else 1.0,
            traffic_bytes_per_point=(fa[dom]['FETCH_SIZE'] * 1024 * (corr if corr else 1.0) + wa[dom]['WRITE_SIZE'] * 1024) / proofs / points,
            traffic_over_algorithmic_bytes=(fa[dom]['FETCH_SIZE'] * 1024 * (corr if corr else 1.0) + wa[dom]['WRITE_SIZE'] * 1024) / proofs / points / 96.0,
            note='WRITE_SIZE is exact (XYZZ buckets of 128 B: W*B per launch, B in the merged form).  FETCH_SIZE is reported RAW: the guide\'s x2 gfx950 '
                 'correction is calibrated for wide coalesced streams (it holds for ntt_pass_kernel in this same pass), while this '
                 'kernel gathers 64-byte points at random 64-B-aligned addresses -- an uncalibrated width.  Expected demand: W (12-13) windows x '
                 '(64 B point + 4 B index), about 850 B per non-trivial point.  Traffic is several times the 96 B/point algorithmic bytes '
                 'because Pippenger re-gathers every base once per window; the rate stays far below HBM peak: the kernel is VALU-bound.'))
    json.dump(j, open(out, 'w'), indent=1)
    print('wrote', out, 'proofs in the pass', proofs, 'dominant fetch GB per proof', fa[dom]['FETCH_SIZE'] * 1024 / 1e9 / proofs)


def valu(sq_dir, derived_dir, out):
    sa, sl = per_kernel(load(sq_dir))
    da, dl = per_kernel(load(derived_dir))
    res = {}
    for k in sorted(sa, key=lambda k: -sa[k].get('SQ_BUSY_CYCLES', 0))[:16]:
        e = dict(launches=sl[k])
        e.update({c: v for c, v in sa[k].items()})
        for c in ('VALUBusy', 'VALUUtilization', 'MemUnitBusy'):
            if k in da and c in da[k]:
                e[c + '_avg'] = da[k][c] / dl[k]
        wc = sa[k].get('SQ_WAVE_CYCLES', 0)
        if wc:
            e['valu_active_share_of_wave_cycles'] = sa[k].get('SQ_ACTIVE_INST_VALU', 0) / wc
            e['wait_any_share_of_wave_cycles'] = sa[k].get('SQ_WAIT_ANY', 0) / wc
            e['wait_inst_share_of_wave_cycles'] = sa[k].get('SQ_WAIT_INST_ANY', 0) / wc
        res[k] = e
    json.dump(dict(_doc='rocprofv3 --pmc passes over `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline` (2^25 rows). Pass 1: SQ_BUSY_CYCLES '
                        'SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE; pass 2: derived '
                        'VALUBusy VALUUtilization MemUnitBusy.  Sums over launches; *_avg are per-launch means.', kernels=res), open(out, 'w'), indent=1)
    print('wrote', out)


if __name__ == '__main__':
    if sys.argv[1] == 'traffic':
        traffic(sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6],
                int(sys.argv[7]) if len(sys.argv) > 7 and sys.argv[7] != 'auto' else None,
                sys.argv[8] if len(sys.argv) > 8 else 'synthetic', sys.argv[9] if len(sys.argv) > 9 and sys.argv[9] != '-' else None,
                int(sys.argv[10]) if len(sys.argv) > 10 else None)
    else:
        valu(sys.argv[2], sys.argv[3], sys.argv[4])
