#!/usr/bin/env python3
"""Per-rank compute budget of the N-GPU prover, measured on ONE GPU: for W = 1, 2, 4, 8 the share of rank 0 of W -- its key
shard, its cyclic row slice, its 1/W of every transform, its slices of the five multiplications -- runs ALONE on the device
through the same C-ABI pieces the multi-GPU schedules issue (fk_r1cs_eval_slice_dev, fk_dq_*, fk_prove_msms_hz_r1cs_dev), with
the all-to-all replaced by a device-local copy of the same size.  What a rank's GPU has to do per proof is therefore measured;
what it has to wait for -- the xGMI exchanges -- is link arithmetic (printed beside it), since a one-GPU box has no links.

    python tools/rank_budget.py [--copies 1024] [--ranks 1,2,4,8]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402  (workload helpers only)
import fawkes_crypto_amd as fk  # noqa: E402
from fawkes_crypto_amd import parallel  # noqa: E402

XGMI_GBPS_PER_LINK_PER_DIRECTION = 64.0      # 128 GB/s bidirectional per link (MI355X_MICROARCH.md: 7 links x ~153 GB/s peak); ~64 GB/s one way in practice


class Buf:
    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes, self.ptr = ctx, nbytes, ctx.dev_alloc(nbytes)

    def data_ptr(self):
        return self.ptr

    def free(self):
        self.ctx.dev_free(self.ptr)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--copies', type=int, default=1024)
    ap.add_argument('--ranks', default='1,2,4,8')
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--split', choices=('work', 'equal', 'q0'), default='work', help="how l, a, b_g1, b_g2 are dealt to the ranks (FK_Z_WORK_SPLIT / FK_Z_EQUAL_SPLIT)")
    ap.add_argument('--all-ranks', action='store_true', help='measure every rank (default: rank 0 only; with the work split the ranks differ)')
    args = ap.parse_args()
    ctx = fk.Context(0)
    r1cs, zs = bench.load_rollup_instance()
    copies = args.copies
    num_input = 1 + copies * (r1cs.num_input - 1)
    n = copies * r1cs.num_gates + num_input
    log_m = max(n - 1, 1).bit_length()
    m = 1 << log_m
    z = bench.tile_witness(zs, r1cs.num_input, copies)
    d_z = ctx.dev_alloc(z.nbytes)
    ctx.upload(d_z, z)
    dr = ctx.load_r1cs(r1cs, copies=copies)
    tox = {k: bench.mont(v) for k, v in bench.TOXIC.items()}
    out = {'workload': '%d rollup-style transactions, domain 2^%d' % (copies, log_m), 'ranks': {}}
    split = {'work': fk.api.Z_WORK_SPLIT, 'equal': fk.api.Z_EQUAL_SPLIT, 'q0': fk.api.Z_WORK_SPLIT_Q0}[args.split]
    q0 = args.split == 'q0'          # the exchange-free schedule: rank 0 evaluates, computes the WHOLE quotient and H; the others run witness multiplications only
    for W in [int(x) for x in args.ranks.split(',')]:
        lw = 0 if q0 else parallel.log2_world(W)
        L = m >> lw
        send = [Buf(ctx, L * 32) for _ in range(3)]
        recv = [Buf(ctx, L * 32) for _ in range(3)]

        def a2a(dst, src):          # the same bytes moved, device-locally (the exchange itself is not what is measured here)
            for d_, s_ in zip(dst, src):
                ctx.dev_copy(d_.data_ptr(), s_.data_ptr(), s_.nbytes)

        def timed(fn):
            fn(); ctx.sync()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                fn()
            ctx.sync()
            return (time.perf_counter() - t0) / args.reps * 1e3

        # with the arrays dealt by work the ranks hold different pieces: every rank's share is measured, the slowest one counts
        ranks = range(W) if (args.all_ranks and W > 1) else [0]
        per_rank = []
        for g in ranks:
            key, _ = ctx.setup(r1cs, copies=copies, shard_index=g, shard_count=W, z_frac=split if W > 1 else fk.api.Z_EQUAL_SPLIT, **tox)

            def ev():
                ctx.r1cs_eval_slice_dev(dr, d_z, log_m, g, lw, *[b.data_ptr() for b in send])

            def quot():
                return parallel.quotient_distributed(ctx, g, W, None, n, log_m, send, recv, a2a, slices_in_send=True)

            def msms():
                return ctx.prove_msms_hz_r1cs_dev(key, dr, send[0].data_ptr(), d_z)

            def whole():
                ev(); blk = quot()
                return ctx.prove_msms_hz_r1cs_dev(key, dr, blk.data_ptr(), d_z)

            if q0 and W > 1:
                if g == 0:
                    full = [Buf(ctx, m * 32) for _ in range(4)]
                    ev = lambda: ctx.r1cs_eval_dev(dr, d_z, full[0].data_ptr(), full[1].data_ptr(), full[2].data_ptr())
                    quot = lambda: ctx.quotient_h_dev(full[0].data_ptr(), full[1].data_ptr(), full[2].data_ptr(), n, full[3].data_ptr())
                    msms = lambda: ctx.prove_msms_hz_r1cs_dev(key, dr, full[3].data_ptr(), d_z)
                    whole = lambda: (ev(), quot(), msms())
                else:
                    full = []
                    ev = quot = lambda: None
                    msms = whole = lambda: ctx.prove_msms_hz_r1cs_dev(key, dr, 0, d_z)
            t_ev = timed(ev)
            ev(); t_q = timed(quot)
            t_m = timed(msms)
            t_all = timed(whole)
            if q0 and W > 1:
                for b_ in full:
                    b_.free()
            info = key.shard_info()
            per_rank.append({'rank': g, 'eval_slice_ms': t_ev, 'quotient_compute_ms': t_q, 'msms_ms': t_m, 'whole_share_ms': t_all, 'levels': key.precomputed(),
                             'points': {k_: v_[1] - v_[0] for k_, v_ in info.items()}})
            key.free()
        worst = max(per_rank, key=lambda e: e['whole_share_ms'])
        # exchanges: 7 per proof, each rank sends (W - 1) / W of its L * 32 bytes, one chunk per peer, every peer on a link of its own
        chunk = L * 32 // W
        t_x = 7 * (chunk / (XGMI_GBPS_PER_LINK_PER_DIRECTION * 1e9)) * 1e3 if (W > 1 and not q0) else 0.0
        # the witness (round 5): a rank uploads ceil(nv / W) elements over its own PCIe link (~50 GB/s pinned) and collects the other W - 1
        # pieces from its peers, one piece per link, all links at once -- both underneath the PREVIOUS proof, so they bound the rate only
        # where they exceed a rank's compute share.  (Rounds 3-4: every rank uploaded all of z out of the same pinned host buffer.)
        piece = -(-z.shape[0] // W) * 32
        t_up = piece / 50e9 * 1e3
        t_ag = (piece / (XGMI_GBPS_PER_LINK_PER_DIRECTION * 1e9)) * 1e3 if W > 1 else 0.0
        out['ranks'][str(W)] = dict(worst, split=args.split if W > 1 else None, per_rank=per_rank, witness_upload_bytes_per_rank=int(piece),
                                    witness_upload_bytes_per_rank_rounds_3_4=int(z.nbytes), witness_all_gather_bytes_into_each_rank=int(piece * (W - 1)),
                                    witness_upload_ms_at_50GBps=t_up, witness_all_gather_ms_by_link_arithmetic=t_ag,
                                    host_read_GBps_for_uploads_at_this_rate=(z.nbytes / 1e9) / (worst['whole_share_ms'] / 1e3),
                                    host_read_GBps_rounds_3_4=(W * z.nbytes / 1e9) / (worst['whole_share_ms'] / 1e3),
                                    exchange_bytes_per_rank_per_proof=7 * chunk * (W - 1), exchange_ms_by_link_arithmetic=t_x)
        print('W = %d (%s split): slowest rank %d: eval %.2f ms, quotient %.2f ms, MSMs %.2f ms, whole share %.2f ms; exchanges %.2f ms by link arithmetic'
              % (W, args.split if W > 1 else '-', worst['rank'], worst['eval_slice_ms'], worst['quotient_compute_ms'], worst['msms_ms'], worst['whole_share_ms'], t_x), flush=True)
        print('        witness: %.0f MB per rank over PCIe (%.1f ms at 50 GB/s; rounds 3-4: %.0f MB), all-gather of %.0f MB into every rank (%.1f ms by link arithmetic), '
              'both underneath the previous proof; host buffer read at %.0f GB/s in all (rounds 3-4: %.0f GB/s)'
              % (piece / 1e6, t_up, z.nbytes / 1e6, piece * (W - 1) / 1e6, t_ag, (z.nbytes / 1e9) / (worst['whole_share_ms'] / 1e3), (W * z.nbytes / 1e9) / (worst['whole_share_ms'] / 1e3)), flush=True)
        if len(per_rank) > 1:
            print('        per rank (whole share ms / points l, a, b_g1, b_g2): ' + '; '.join(
                '%d: %.1f / %s' % (e['rank'], e['whole_share_ms'], ','.join('%.1fM' % (e['points'][k_] / 1e6) for k_ in ('l', 'a', 'b', 'b_g2'))) for e in per_rank), flush=True)
        for b in send + recv:
            b.free()
    print(json.dumps(out))


if __name__ == '__main__':
    main()
