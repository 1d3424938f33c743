#!/usr/bin/env python3
"""Robustness probe (SURVEY 8(b) "Errors": status codes, never aborts): the N-rank prover with almost no free HBM.  A dummy allocation
leaves `--free-gb` of device memory; keys are set up and proofs attempted on 8 ranks sharing the GPU.  Every outcome must be a proof
or an FkError (FK_ERR_OOM) -- never a crash -- and after the dummy is released the same context must prove correctly.

    FK_DEBUG=1 python3 tools/oom_probe.py [--copies 256] [--free-gb 6,3,1.5]
"""
import argparse
import faulthandler
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
import fawkes_crypto_amd as fk  # noqa: E402

faulthandler.enable(all_threads=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--copies', type=int, default=256)
    ap.add_argument('--ranks', type=int, default=8)
    ap.add_argument('--free-gb', default='6,3,1.5,0.5')
    args = ap.parse_args()
    import torch
    inst, zs = bench.load_rollup_instance()
    z = bench.tile_witness(zs, inst.num_input, args.copies)
    tox = {k: bench.mont(v) for k, v in bench.TOXIC.items()}
    r, s = bench.mont(0xA11CE), bench.mont(0xB0B)
    ctx = fk.Context(0)
    dr = ctx.load_r1cs(inst, copies=args.copies)
    key, _ = ctx.setup(inst, copies=args.copies, **tox)
    want = ctx.prove_witness(key, dr, z, r, s).tobytes()
    key.free(); dr.free(); ctx.trim()
    for when in ('before the keys are set up', 'after the keys are set up, before the first proof'):
        for gb in [float(x) for x in args.free_gb.split(',')]:
            mc, dummy = None, 0

            def squeeze():
                free, total = torch.cuda.mem_get_info(0)
                take = free - int(gb * (1 << 30))
                d = ctx.dev_alloc(take) if take > 0 else 0
                print('%s: free HBM %.1f GB -> dummy of %.1f GB, %.2f GB left' % (when, free / 2**30, take / 2**30, torch.cuda.mem_get_info(0)[0] / 2**30), flush=True)
                return d
            try:
                if when.startswith('before'):
                    dummy = squeeze()
                mc = fk.MultiContext([0] * args.ranks)
                mkey, _ = mc.setup(inst, copies=args.copies, **tox)
                mdr = mc.load_r1cs(inst, copies=args.copies)
                if not when.startswith('before'):
                    dummy = squeeze()
                for attempt in range(2):
                    try:
                        got = mc.prove_witness(mkey, mdr, z, r, s).tobytes()
                        print('    attempt %d: proof %s; levels of rank 0: %s' % (attempt, 'OK' if got == want else 'WRONG', mc.key_shard(mkey, 0).precomputed()), flush=True)
                        assert got == want
                    except fk.FkError as e:
                        print('    attempt %d: FkError %s' % (attempt, str(e)[:200]), flush=True)
                    if dummy and attempt == 0:
                        ctx.dev_free(dummy); dummy = 0          # the second attempt has the memory: the failed call must have left every rank usable
            except fk.FkError as e:
                print('    FkError %s' % str(e)[:200], flush=True)
            finally:
                if mc is not None:
                    mc.close()
                if dummy:
                    ctx.dev_free(dummy)
    # and afterwards everything works
    dr = ctx.load_r1cs(inst, copies=args.copies)
    key, _ = ctx.setup(inst, copies=args.copies, **tox)
    assert ctx.prove_witness(key, dr, z, r, s).tobytes() == want
    print('after the squeeze: single-GPU proof OK', flush=True)


if __name__ == '__main__':
    main()
