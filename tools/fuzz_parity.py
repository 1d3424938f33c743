#!/usr/bin/env python3
"""Differential fuzzing of the HIP library against the CPU oracle (GPU box; the oracle is the checker, as in tests/).

Seeded random cases for a bounded time: multi-scalar multiplications with hostile scalar and base mixes, transforms and
quotients of every small size, and whole proofs of random constraint systems -- ragged rows (empty, single, hundreds of
terms), repeated columns, zero / one / r-1 coefficients, unsatisfied systems (the prover's bytes are defined for any
witness: h is the quotient, the remainder is dropped) -- through randomly chosen entry points: oracle key or GPU key
generation or a Parameters file, host or device evaluation, merged fixed-base levels or not, one context or the in-library
multi-rank form on 1-4 ranks, single calls or pipelined tickets.  Every result must equal the oracle's bytes.

    python tools/fuzz_parity.py [--seconds 600] [--seed 1]        # prints one line per case kind and a summary; rc != 0 on any mismatch
"""
import argparse
import os
import random
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

import bn254_ref as ref  # noqa: E402
import c_oracle as co  # noqa: E402
import fixtures as fx  # noqa: E402
import fawkes_crypto_amd as fk  # noqa: E402
from helpers import g1_bases, g2_bases, params_from_oracle_key, r1cs_product  # noqa: E402

R = ref.R
MONT = ref.MONT_R % R
STATS = {}


def mont_arr(vals):
    return co.limbs_arr([v % R * MONT % R for v in vals]) if len(vals) else np.zeros((0, 4), np.uint64)


def scalar(rnd):
    k = rnd.random()
    if k < 0.15: return 0
    if k < 0.35: return 1
    if k < 0.40: return R - 1
    if k < 0.45: return R - 1 - rnd.randrange(4)
    if k < 0.55: return rnd.randrange(1, 1 << rnd.randrange(1, 64))
    if k < 0.62: return 1 << rnd.randrange(0, 254)
    if k < 0.66: return (1 << rnd.randrange(1, 254)) - 1
    return rnd.getrandbits(256) % R


def case_msm(ctx, rnd, g2):
    n = rnd.choice([1, 2, 3, 63, 64, 65, rnd.randrange(1, 600), rnd.randrange(600, 4000)]) if not g2 else rnd.choice([1, 2, 5, 64, rnd.randrange(1, 500)])
    if not g2 and rnd.random() < 0.08:
        n = rnd.randrange(20000, 70000)
    pool = (g2_bases if g2 else g1_bases)(max(1, n // rnd.choice([1, 1, 2, 7])), seed=rnd.randrange(1000))
    idx = [rnd.randrange(len(pool)) for _ in range(n)]
    bases = pool[idx].copy()
    w = 128 if g2 else 64
    dec, enc, neg = (ref.g2_from_raw_le, ref.g2_raw_le, ref.G2.neg) if g2 else (ref.g1_from_raw_le, ref.g1_raw_le, ref.G1.neg)
    for i in range(n):
        k = rnd.random()
        if k < 0.03:
            bases[i] = 0                                     # the identity
        elif k < 0.08:
            bases[i] = np.frombuffer(enc(neg(dec(bases[i].tobytes()))), np.uint8)      # -P of a pool point: cancellations inside buckets
    mode = rnd.random()
    if mode < 0.2:
        one = scalar(rnd)
        sc = [one] * n                                       # every digit the same: one bucket per window, the oversized path
    elif mode < 0.4:
        few = [scalar(rnd) for _ in range(3)]
        sc = [rnd.choice(few) for _ in range(n)]
    else:
        sc = [scalar(rnd) for _ in range(n)]
    sm = mont_arr(sc)
    assert bases.shape == (n, w)
    c_bits = rnd.choice([0, 0, rnd.randrange(2, 23)])        # 0 = the library's choice
    ctx.set_window_bits(c_bits)
    try:
        got = (ctx.msm_g2 if g2 else ctx.msm_g1)(bases, sm).tobytes()
    finally:
        ctx.set_window_bits(0)
    want = (co.msm_g2 if g2 else co.msm_g1)(bases, sm).tobytes()
    assert got == want, 'msm mismatch (c = %d)' % c_bits
    return n


def case_ntt(ctx, rnd):
    log_n = rnd.randrange(0, 15)
    n = 1 << log_n
    data = mont_arr([scalar(rnd) for _ in range(n)])
    inv, coset = rnd.random() < 0.5, rnd.random() < 0.5
    assert np.array_equal(ctx.ntt(data, inverse=inv, coset=coset), co.fr_ntt(data, inverse=inv, coset=coset)), 'ntt mismatch'
    return n


def case_quotient(ctx, rnd):
    n = rnd.choice([1, 2, 3, 5, 64, 65, rnd.randrange(1, 3000), rnd.randrange(3000, 20000)])
    a, b, c = (mont_arr([scalar(rnd) for _ in range(n)]) for _ in range(3))
    assert np.array_equal(ctx.quotient_h(a, b, c), co.quotient_h(a, b, c)), 'quotient mismatch'
    return n


def rand_matrix(rnd, rows, nv, heavy):
    ptr, col, val = [0], [], []
    for _ in range(rows):
        k = rnd.random()
        ln = 0 if k < 0.08 else 1 if k < 0.5 else rnd.randrange(2, 5) if k < 0.88 else rnd.randrange(5, 40) if k < 0.99 or not heavy else rnd.randrange(100, 600)
        for _ in range(ln):
            col.append(rnd.randrange(nv) if rnd.random() < 0.95 else (col[-1] if col else 0))       # a repeated column now and then
            k = rnd.random()
            val.append(1 if k < 0.35 else 0 if k < 0.353 else R - 1 if k < 0.43 else rnd.randrange(1, 1 << 16) if k < 0.5 else rnd.getrandbits(256) % R)
        ptr.append(len(col))
    return co.Csr(np.array(ptr, np.uint64), np.array(col, np.uint32), mont_arr(val))


def rand_system(rnd):
    gates = rnd.choice([1, 2, 7, rnd.randrange(1, 300), rnd.randrange(300, 2500), rnd.randrange(300, 2500)])
    if rnd.random() < 0.07:
        gates = rnd.randrange(8000, 30000)
    nin = rnd.choice([1, 1, 2, 3, rnd.randrange(1, 20)])
    naux = rnd.choice([1, 2, max(1, gates // 2), gates, gates + rnd.randrange(1, 50)])
    nv = nin + naux
    heavy = rnd.random() < 0.3
    A, B = rand_matrix(rnd, gates, nv, heavy), rand_matrix(rnd, gates, nv, heavy)
    zs = [1] + [scalar(rnd) for _ in range(nv - 1)]
    z = mont_arr(zs)
    if rnd.random() < 0.5:
        Cm = rand_matrix(rnd, gates, nv, False)               # unsatisfied (almost surely): the prover's bytes are defined all the same
    else:                                                     # satisfied: row i of C is (a_i * b_i) * ONE
        seq = np.arange(gates + 1, dtype=np.uint64)
        tmp = co.R1csC(nin, naux, A, B, co.Csr(np.zeros(gates + 1, np.uint64), np.zeros(0, np.uint32), np.zeros((0, 4), np.uint64)))
        a, b, *_ = co.synthesize(tmp, z)
        Cm = co.Csr(seq, np.zeros(gates, np.uint32), co.fe_mul_batch(co.FR, a[:gates], b[:gates]))
    return co.R1csC(nin, naux, A, B, Cm), z, zs[:nin]


def case_prove(ctx, rnd, stats):
    cs, z, z_in = rand_system(rnd)
    tox = {k: rnd.randrange(1, R) for k in ('tau', 'alpha', 'beta', 'gamma', 'delta')}
    key = co.setup(cs, **tox)
    a, b, c, aa, bi, ba = co.synthesize(cs, z)
    r, s = fx.mont_fr(scalar(rnd)), fx.mont_fr(scalar(rnd))
    try:
        want = co.prove(key, a, b, c, z, aa, bi, ba, r, s).tobytes()
    except RuntimeError as e:
        # a variable that appears (density bit set) in a column that evaluates to zero (zero coefficients, +k and -k on a
        # repeated column): bellman's key generation drops the point, its prover then runs out of bases -> an error there,
        # rc -3 in the oracle, and an error -- not bytes -- from every product entry point
        assert 'rc=-3' in str(e), e
        want = None
        stats['inconsistent'] = stats.get('inconsistent', 0) + 1
    params = params_from_oracle_key(key, r1cs_product(cs))
    toxm = {k: fx.mont_fr(v) for k, v in tox.items()}
    if rnd.random() < 0.5:
        os.environ['FK_MSM_PRE_MIN_LOG2'] = '6'              # merged fixed-base levels on small keys
    else:
        os.environ.pop('FK_MSM_PRE_MIN_LOG2', None)
    variants = rnd.sample(['oracle_key', 'gpu_setup', 'raw', 'multi', 'tickets', 'file', 'shards', 'verify'], 3)
    for v in variants:
        stats[v] = stats.get(v, 0) + 1
        if want is None:
            try:
                got = prove_variant(ctx, rnd, v, cs, key, params, toxm, z, r, s, (a, b, c, aa, bi, ba), None)
            except fk.FkError:
                continue
            raise AssertionError('inconsistent key / density accepted (%s)' % v)
        got = prove_variant(ctx, rnd, v, cs, key, params, toxm, z, r, s, (a, b, c, aa, bi, ba), want)
        assert got == want, 'proof mismatch (%s)' % v
    del key
    return cs.num_gates


def prove_variant(ctx, rnd, v, cs, key, params, toxm, z, r, s, abc, want):
    a, b, c, aa, bi, ba = abc
    if True:
        if v == 'oracle_key':
            dk = ctx.load_key(params); dr = ctx.load_r1cs(params.r1cs)
            got = ctx.prove_witness(dk, dr, z, r, s).tobytes()
            dr.free(); dk.free()
        elif v == 'gpu_setup':
            dk, _ = ctx.setup(params.r1cs, **toxm); dr = ctx.load_r1cs(params.r1cs)
            for name in ('h', 'l', 'a', 'b_g1', 'b_g2'):
                assert dk.download(name).tobytes() == np.array(getattr(key, name)).tobytes(), 'setup array %s' % name
            got = ctx.prove_witness(dk, dr, z, r, s).tobytes()
            dr.free(); dk.free()
        elif v == 'raw':
            ga = ctx.synthesize(params.r1cs, z)
            for x, y in zip(ga, (a, b, c, aa, bi, ba)):
                assert np.array_equal(np.asarray(x), np.asarray(y)), 'synthesize'
            dk = ctx.load_key(params)
            got = ctx.prove_raw(dk, a, b, c, z, aa, bi, ba, r, s).tobytes()
            dk.free()
        elif v == 'multi':
            W = rnd.choice([1, 2, 3, 4])
            if W == 1 or rnd.random() < 0.5:
                os.environ['FK_MULTI_FORCE_EXCHANGE'] = '1'
            mc = fk.MultiContext([0] * W)
            mk = mc.setup(params.r1cs, **toxm)[0] if rnd.random() < 0.5 else mc.load_key(params)
            mr = mc.load_r1cs(params.r1cs)
            t1 = mc.prove_witness_submit(mk, mr, z, r, s)
            t2 = mc.prove_witness_submit(mk, mr, z, s, r)
            try:
                got = mc.prove_witness_wait(t1).tobytes()
            finally:
                try:
                    got2 = mc.prove_witness_wait(t2).tobytes()
                except fk.FkError:
                    if want is not None:
                        raise
            assert got2 == co.prove(key, a, b, c, z, aa, bi, ba, s, r).tobytes(), 'multi second ticket'
            mr.free(); mk.free(); mc.close()
            os.environ.pop('FK_MULTI_FORCE_EXCHANGE', None)
        elif v == 'tickets':
            dk = ctx.load_key(params); dr = ctx.load_r1cs(params.r1cs)
            z2 = z.copy()
            if len(z2) > 1:
                z2[-1] = fx.mont_fr(scalar(rnd))
            t1 = ctx.prove_witness_submit(dk, dr, z, r, s)
            t2 = ctx.prove_witness_submit(dk, dr, z2, r, s)
            try:
                got = ctx.prove_witness_wait(t1).tobytes()
            finally:                                            # a failed ticket does not cancel the other one: it is still the caller's to collect
                try:
                    got2 = ctx.prove_witness_wait(t2).tobytes()
                except fk.FkError:
                    if want is not None:
                        raise
            a2, b2, c2, *_ = co.synthesize(cs, z2)
            assert got2 == co.prove(key, a2, b2, c2, z2, aa, bi, ba, r, s).tobytes(), 'second ticket'
            dr.free(); dk.free()
        elif v == 'shards':                                     # the MSM work cut over W key shards, the 384-byte partial results folded on the host
            W = rnd.choice([2, 3, 5, 8])
            dk = ctx.load_key(params)
            parts = []
            for i in range(W):
                sk = ctx.load_key(params, shard_index=i, shard_count=W) if rnd.random() < 0.5 else ctx.setup(params.r1cs, shard_index=i, shard_count=W, **toxm)[0]
                try:
                    parts.append(ctx.prove_msms(sk, a, b, c, z, aa, bi, ba))
                finally:
                    sk.free()
            got = ctx.prove_assemble(dk, np.stack(parts), r, s).tobytes()
            dk.free()
        elif v == 'verify':                                     # the product's own verifier on the product's proof; then on a damaged one
            dk = ctx.load_key(params); dr = ctx.load_r1cs(params.r1cs)
            got = ctx.prove_witness(dk, dr, z, r, s).tobytes()
            dr.free(); dk.free()
            if want is not None:
                vkb = fk.vk_to_borsh(dict(alpha_g1=key.alpha_g1, beta_g2=key.beta_g2, gamma_g2=key.gamma_g2, delta_g2=key.delta_g2, ic=np.array(key.ic)))
                inputs = z[1:cs.num_input]
                sat = bool(np.array_equal(co.fe_mul_batch(co.FR, a, b), c))       # satisfied <=> the proof must verify (soundness the other way)
                STATS['satisfied' if sat else 'unsatisfied'] = STATS.get('satisfied' if sat else 'unsatisfied', 0) + 1
                assert fk.verify(vkb, inputs, got, ctx) == sat, 'verifier: expected %s' % sat
                batch = np.frombuffer(got * 3, np.uint8).reshape(3, 256).copy()
                batch[1, rnd.randrange(256)] ^= 1 << rnd.randrange(8)
                res = fk.verify_batch(ctx, vkb, np.tile(inputs, (3, 1, 1)) if len(inputs) else np.zeros((3, 0, 4), np.uint64), batch)
                assert bool(res[0]) == sat and bool(res[2]) == sat and not bool(res[1]), 'batch verifier: %s (satisfied: %s)' % (list(res), sat)
                try:
                    assert not fk.verify(vkb, inputs, batch[1].tobytes(), ctx), 'damaged proof accepted'
                except fk.FkError:
                    pass                                        # a coordinate >= q: refused as malformed
        else:                                                   # a Parameters file, written by the host mirror and read back on the GPU
            from fawkes_crypto_amd import params_io
            arrays = dict(alpha_g1=key.alpha_g1, beta_g1=key.beta_g1, beta_g2=key.beta_g2, gamma_g2=key.gamma_g2, delta_g1=key.delta_g1,
                          delta_g2=key.delta_g2, ic=np.array(key.ic), h=np.array(key.h), l=np.array(key.l), a=np.array(key.a),
                          b_g1=np.array(key.b_g1), b_g2=np.array(key.b_g2))
            bits = [rnd.random() < 0.5 for _ in range(rnd.randrange(0, 9))]
            how = rnd.randrange(3)
            os.environ['FK_HOST_THREADS'] = str(rnd.choice([1, 2, 3, 5, 16]))          # (read per call: the gate codec's thread count)
            if how == 0:                # the raw stream, written by the per-term Python restatement
                data = params_io.store_parameters(arrays, params.r1cs, const_tracker_bits=bits)
            else:                       # the native encoder (round 5): brotli at a random setting, or its raw stream
                gb = fk.api.GateBlob(params.r1cs, None, fmt=fk.api.FK_GATES_BROTLI if how == 1 else fk.api.FK_GATES_RAW, quality=rnd.choice([0, 1, 5, 9, 11]), lgwin=rnd.choice([10, 16, 22, 24]))
                blob = gb.data.tobytes() if how == 1 else params_io.RAW_MAGIC + gb.data.tobytes()
                gb.free()
                data = params_io.write_parameters(params.r1cs.num_gates, blob, bits, params_io.encode_bellman_parameters(arrays))
            dk, dr, _ = params_io.load_parameters(ctx, data, shard_index=0, shard_count=1, overlap=rnd.random() < 0.7)
            del os.environ['FK_HOST_THREADS']
            got = ctx.prove_witness(dk, dr, z, r, s).tobytes()
            dr.free(); dk.free()
    return got


def case_tiled(ctx, rnd, stats):
    """a batch circuit: `copies` instances of one system, resident once (fk_setup_tiled / fk_r1cs_load_tiled), against the oracle's
    proof of the explicitly replicated system"""
    while True:
        base, _, _ = rand_system(rnd)
        if base.num_gates <= 1500:
            break
    copies = rnd.choice([1, 2, 3, 5, 8, rnd.randrange(2, 40), rnd.randrange(64, 200)])       # 64 and more: the wave form of the evaluation
    if copies >= 64:
        while base.num_gates > 200:
            base, _, _ = rand_system(rnd)
    cs = fx.tile_r1cs(base, copies)
    nv = cs.num_input + cs.num_aux
    z = mont_arr([1] + [scalar(rnd) for _ in range(nv - 1)])
    tox = {k: rnd.randrange(1, R) for k in ('tau', 'alpha', 'beta', 'gamma', 'delta')}
    toxm = {k: fx.mont_fr(v) for k, v in tox.items()}
    key = co.setup(cs, **tox)
    a, b, c, aa, bi, ba = co.synthesize(cs, z)
    r, s = fx.mont_fr(scalar(rnd)), fx.mont_fr(scalar(rnd))
    try:
        want = co.prove(key, a, b, c, z, aa, bi, ba, r, s).tobytes()
    except RuntimeError as e:
        assert 'rc=-3' in str(e), e
        want = None
        stats['inconsistent'] = stats.get('inconsistent', 0) + 1
    if rnd.random() < 0.5:
        os.environ['FK_MSM_PRE_MIN_LOG2'] = '6'
    else:
        os.environ.pop('FK_MSM_PRE_MIN_LOG2', None)
    inst = r1cs_product(base)

    def single():
        dk, _ = ctx.setup(inst, copies=copies, **toxm); dr = ctx.load_r1cs(inst, copies=copies)
        try:
            for name in ('h', 'l', 'a', 'b_g1', 'b_g2'):
                assert dk.download(name).tobytes() == np.array(getattr(key, name)).tobytes(), 'tiled setup array %s' % name
            return ctx.prove_witness(dk, dr, z, r, s).tobytes()
        finally:
            dr.free(); dk.free()

    def multi():
        W = rnd.choice([2, 3, 4])
        mc = fk.MultiContext([0] * W)
        held = []
        try:
            held.append(mc.setup(inst, copies=copies, **toxm)[0]); held.append(mc.load_r1cs(inst, copies=copies))
            return mc.prove_witness(held[0], held[1], z, r, s).tobytes()
        finally:
            for h in held:
                h.free()
            mc.close()

    for name, fn in (('tiled', single), ('tiled_multi', multi)):
        stats[name] = stats.get(name, 0) + 1
        if want is None:
            try:
                fn()
            except fk.FkError:
                continue
            raise AssertionError('inconsistent key / density accepted (%s)' % name)
        assert fn() == want, 'proof mismatch (%s, %d copies)' % (name, copies)
    return cs.num_gates


def case_corrupt(ctx, rnd, stats):
    """a damaged Parameters file: truncated, bytes flipped (in the counts, the points, the gate blob), regions swapped.  The loader
    must refuse it or -- when the damage leaves a decodable file -- load it; it must never crash, hang or leak the device"""
    from fawkes_crypto_amd import params_io
    while True:
        cs, z, _ = rand_system(rnd)
        if cs.num_gates <= 400:
            break
    key = co.setup(cs, **{k: rnd.randrange(1, R) for k in ('tau', 'alpha', 'beta', 'gamma', 'delta')})
    arrays = dict(alpha_g1=key.alpha_g1, beta_g1=key.beta_g1, beta_g2=key.beta_g2, gamma_g2=key.gamma_g2, delta_g1=key.delta_g1,
                  delta_g2=key.delta_g2, ic=np.array(key.ic), h=np.array(key.h), l=np.array(key.l), a=np.array(key.a),
                  b_g1=np.array(key.b_g1), b_g2=np.array(key.b_g2))
    good = params_io.store_parameters(arrays, r1cs_product(cs), const_tracker_bits=[True, False])
    for _ in range(12):
        data = bytearray(good)
        k = rnd.random()
        if k < 0.25:
            data = data[:rnd.randrange(0, len(data))]
        elif k < 0.7:
            for _ in range(rnd.choice([1, 1, 2, 8])):
                data[rnd.randrange(len(data))] ^= 1 << rnd.randrange(8)
        elif k < 0.85:
            i = rnd.randrange(len(data)); data[i:i + 4] = rnd.getrandbits(32).to_bytes(4, 'big')      # a forged count, most likely
        else:
            i, j = sorted(rnd.randrange(len(data)) for _ in range(2)); data = data[:i] + data[j:] + data[i:j]
        try:
            dk, dr, _ = params_io.load_parameters(ctx, bytes(data), checked=rnd.random() < 0.8, disallow_points_at_infinity=rnd.random() < 0.3)
            stats['corrupt_loaded'] = stats.get('corrupt_loaded', 0) + 1
            try:
                if dk.counts()['num_input'] + dk.counts()['num_aux'] == len(z):
                    ctx.prove_witness(dk, dr, z, fx.mont_fr(3), fx.mont_fr(4))      # whatever it proves, it returns
            except fk.FkError:
                pass
            dr.free(); dk.free()
        except (fk.FkError, ValueError, AssertionError, IndexError, OverflowError, MemoryError) as e:      # refused: by the library or by the host-side reader
            stats['corrupt_refused'] = stats.get('corrupt_refused', 0) + 1
            stats.setdefault('corrupt_kinds', set()).add(type(e).__name__)
    return 12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=600)
    ap.add_argument('--seed', type=int, default=1)
    args = ap.parse_args()
    co.build(); co.lib()
    ctx = fk.Context(0)
    kinds = [('msm_g1', 3), ('msm_g2', 2), ('ntt', 2), ('quotient', 2), ('prove', 6), ('tiled', 2), ('corrupt', 1)]
    bag = [k for k, w in kinds for _ in range(w)]
    counts, units, fails, stats = {}, {}, [], STATS
    t0 = time.time()
    i = 0
    while time.time() - t0 < args.seconds:
        case_seed = args.seed * 1000003 + i
        rnd = random.Random(case_seed)
        kind = rnd.choice(bag)
        try:
            if kind == 'msm_g1': u = case_msm(ctx, rnd, False)
            elif kind == 'msm_g2': u = case_msm(ctx, rnd, True)
            elif kind == 'ntt': u = case_ntt(ctx, rnd)
            elif kind == 'quotient': u = case_quotient(ctx, rnd)
            elif kind == 'tiled': u = case_tiled(ctx, rnd, stats)
            elif kind == 'corrupt': u = case_corrupt(ctx, rnd, stats)
            else: u = case_prove(ctx, rnd, stats)
            counts[kind] = counts.get(kind, 0) + 1
            units[kind] = units.get(kind, 0) + u
        except Exception as e:      # noqa: BLE001 -- a fuzzer reports and goes on
            fails.append((case_seed, kind, repr(e)))
            print('FAIL case_seed=%d kind=%s: %s' % (case_seed, kind, e), flush=True)
            traceback.print_exc()
            if len(fails) >= 10:
                break
        i += 1
    for k, _ in kinds:
        print('%-9s %5d cases, %9d units' % (k, counts.get(k, 0), units.get(k, 0)))
    print('prove variants:', stats)
    print('fuzz: %d cases in %.0f s, seed %d, %d failures' % (i, time.time() - t0, args.seed, len(fails)))
    sys.exit(1 if fails else 0)


if __name__ == '__main__':
    main()
