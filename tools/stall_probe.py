#!/usr/bin/env python3
"""Looks for the sporadic ~300 ms stall seen in the standalone multiplication legs (one of three repetitions of
the 2^25 G1 multiplication took ~400 ms instead of 40 on the driver's box; round 4's first run: one of ten G2 repetitions 387 ms
instead of 72).  Times many consecutive repetitions of (a) the H multiplication alone and (b) the pipelined proof loop, prints every
repetition beyond 1.5x the median with its start time, and optionally runs `rocm-smi` beside them every few seconds.

    python3 tools/stall_probe.py [--copies 1741] [--reps 300] [--proofs 60] [--smi-period 0]
"""
import argparse
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402


def report(name, t_start, times):
    ts = np.array(times)
    med = float(np.median(ts))
    out = [(t_start[i], ts[i]) for i in range(len(ts)) if ts[i] > 1.5 * med]
    print('%s: %d reps, median %.2f ms, mean %.2f ms, max %.2f ms, %d outliers > 1.5 x median' % (name, len(ts), med * 1e3, ts.mean() * 1e3, ts.max() * 1e3, len(out)), flush=True)
    for t0, d in out:
        print('    at t = %7.2f s: %.1f ms' % (t0, d * 1e3), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--copies', type=int, default=1741)
    ap.add_argument('--reps', type=int, default=300)
    ap.add_argument('--proofs', type=int, default=60)
    ap.add_argument('--smi-period', type=float, default=0.0, help='run `rocm-smi --showuse --showmeminfo vram` every this many seconds beside the loops (0 = never)')
    args = ap.parse_args()
    import fawkes_crypto_amd as fk
    ctx = fk.Context(0)
    inst, zs = bench.load_rollup_instance()
    tox = {k: bench.mont(v) for k, v in bench.TOXIC.items()}
    r, s = bench.mont(0xA11CE), bench.mont(0xB0B)
    copies = args.copies
    nv = 1 + copies * (inst.num_input - 1) + copies * inst.num_aux
    z_pin = [ctx.host_alloc((nv, 4)) for _ in range(2)]
    bench.tile_witness(zs, inst.num_input, copies, out=z_pin[0])
    bench.tile_witness(zs[::-1], inst.num_input, copies, out=z_pin[1])
    dr = ctx.load_r1cs(inst, copies=copies)
    key, vk = ctx.setup(inst, copies=copies, **tox)
    t_zero = time.perf_counter()
    stop = threading.Event()

    def smi_loop():
        while not stop.wait(args.smi_period):
            t0 = time.perf_counter()
            subprocess.run(['rocm-smi', '--showuse', '--showmeminfo', 'vram', '--json'], capture_output=True)
            print('    [rocm-smi at t = %.2f s took %.0f ms]' % (t0 - t_zero, (time.perf_counter() - t0) * 1e3), flush=True)

    th = None
    if args.smi_period > 0:
        th = threading.Thread(target=smi_loop, daemon=True)
        th.start()
    info = key.shard_info()
    n_h = info['h'][1] - info['h'][0]
    d_s = ctx.dev_alloc(n_h * 32)
    ctx.gen_scalars_dev(d_s, n_h, 17, 0)
    for _ in range(2):
        ctx.prove_msm_array_dev(key, 'h', d_s)
    t_start, times = [], []
    for _ in range(args.reps):
        ctx.sync()
        t0 = time.perf_counter()
        ctx.prove_msm_array_dev(key, 'h', d_s)
        ctx.sync()
        t_start.append(t0 - t_zero); times.append(time.perf_counter() - t0)
    report('H multiplication alone (%d points)' % n_h, t_start, times)
    ctx.dev_free(d_s)
    # the pipelined loop bench.py times
    tk = ctx.prove_witness_submit(key, dr, z_pin[0], r, s)
    t_start, times = [], []
    for i in range(args.proofs + 2):
        t0 = time.perf_counter()
        nxt = ctx.prove_witness_submit(key, dr, z_pin[(i + 1) & 1], r, s)
        ctx.prove_witness_wait(tk); tk = nxt
        if i >= 2:
            t_start.append(t0 - t_zero); times.append(time.perf_counter() - t0)
    ctx.prove_witness_wait(tk)
    report('pipelined proofs (%d transactions)' % copies, t_start, times)
    stop.set()
    key.free(); dr.free()
    for zp in z_pin:
        ctx.host_free(zp)


if __name__ == '__main__':
    main()
