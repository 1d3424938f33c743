#!/usr/bin/env python3
"""COPIES eddsa-poseidon signature checks (CIRCUIT=eddsa, default) or rollup-style transactions (CIRCUIT=rollup: two depth-32
poseidon merkle proofs + one eddsa signature each, oracle/fawkes_circuit.py: rollup_tx_circuit; COPIES=1024 TILED=1 is the
"1024-tx shape" of BASELINE configs[3] at 2^25) as ONE constraint system (BASELINE configs[2]; COPIES=4096 TILED=1 is its full
size): GPU setup, witness -> proof timing, pairing check.  TILED=1 keeps one instance resident (fk_r1cs_load_tiled /
fk_setup_tiled) instead of the replicated matrices.  Uses the oracle-side circuit builder: a probe, not product code."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np
import fawkes_circuit as fc, fixtures as fx, bn254_ref as ref
from helpers import r1cs_product, TOXIC
copies = int(os.environ.get('COPIES', '256'))
t0 = time.time()
distinct = int(os.environ.get('DISTINCT', '64'))
pp, jj = fc.PoseidonParams(4, 8, 54), fc.JubJubBN256()


circuit = os.environ.get('CIRCUIT', 'eddsa')


def instance(k):
    if circuit == 'rollup':
        import random
        rnd = random.Random(k)
        return fc.rollup_tx_circuit(1000003 * (k + 1), 1000 + k, 900 + k, [rnd.randrange(ref.R) for _ in range(32)], [rnd.randrange(2) for _ in range(32)], 555 + 31 * k)
    return fc.eddsa_circuit(1000003 * (k + 1), 777 + k, 555 + 31 * k, pp, jj)[0]


def sign(k):
    cs = instance(k)
    return cs.z_in[1:], fx.witness_mont(cs.z_in, cs.z_aux)


# the circuit synthesis is Python (about 0.5 s per signature): spread over the host cores, BEFORE anything touches the GPU
workers = min(os.cpu_count() or 1, int(os.environ.get('WORKERS', '64')), distinct)
cache = os.environ.get('ZCACHE')        # e.g. /tmp/eddsa_z.npz: a second (profiled) run on the same box skips the synthesis
if cache and os.path.exists(cache):
    ld = np.load(cache, allow_pickle=True)
    made = list(zip([list(x) for x in ld['pub']], list(ld['zs'])))
    assert len(made) == distinct
elif workers > 1 and distinct > 8:
    import multiprocessing as mp
    with mp.get_context('fork').Pool(workers) as pool:
        made = pool.map(sign, range(distinct), chunksize=max(1, distinct // (4 * workers)))
else:
    made = [sign(k) for k in range(distinct)]
if cache and not os.path.exists(cache):
    np.savez(cache, pub=np.array([m[0] for m in made], dtype=object), zs=np.stack([m[1] for m in made]))
pub, zs = [m[0] for m in made], [m[1] for m in made]
sigs = [instance(0)]
import fawkes_crypto_amd as fk
one = fx.r1cs_to_csr(sigs[0].r1cs())
tiled = os.environ.get('TILED', '0') == '1'
ni = one.num_input
z = np.ascontiguousarray(np.concatenate([zs[0][:1]] + [zs[k % distinct][1:ni] for k in range(copies)] + [zs[k % distinct][ni:] for k in range(copies)]))
terms = sum(len(m.col) for m in (one.A, one.B, one.C))
print('built %d instances (%d distinct, %d host processes): %d gates, %d variables, %d matrix terms in %.1f s' % (copies, distinct, workers, copies * one.num_gates,
      len(z), copies * terms, time.time() - t0), flush=True)
ctx = fk.Context(0)
t0 = time.time()
tox = {k: fx.mont_fr(v) for k, v in TOXIC.items()}
if tiled:
    dk, vk = ctx.setup(r1cs_product(one), copies=copies, **tox)
    dr = ctx.load_r1cs(r1cs_product(one), copies=copies)
else:
    r1cs = r1cs_product(fx.tile_r1cs(one, copies))
    dk, vk = ctx.setup(r1cs, **tox)
    dr = ctx.load_r1cs(r1cs)
print('setup + load %.1f s, domain 2^%d' % (time.time() - t0, dk.counts()['m'].bit_length() - 1), flush=True)
r, s = fx.mont_fr(11), fx.mont_fr(22)
proof = ctx.prove_witness(dk, dr, z, r, s)
t0 = time.time()
for _ in range(5):
    p2 = ctx.prove_witness(dk, dr, z, r, s)
dt = (time.time() - t0) / 5
assert p2.tobytes() == proof.tobytes()
print('%d %s instances per proof: %.2f ms per proof (witness upload included), %.0f instances proved per second' % (copies, circuit, dt * 1e3, copies / dt), flush=True)
d_z = ctx.dev_alloc(z.nbytes); ctx.upload(d_z, z)
ctx.prove_witness_dev(dk, dr, d_z, r, s)
t0 = time.time()
for _ in range(5):
    p3, tm = ctx.prove_witness_dev(dk, dr, d_z, r, s, want_timings=True)
dt = (time.time() - t0) / 5
assert p3.tobytes() == proof.tobytes()
print('witness resident in HBM: %.2f ms per proof, %.0f instances proved per second' % (dt * 1e3, copies / dt), flush=True)
g1 = lambda b: ref.g1_from_raw_le(bytes(b)); g2 = lambda b: ref.g2_from_raw_le(bytes(b))
pk = dict(alpha_g1=g1(vk['alpha_g1']), beta_g2=g2(vk['beta_g2']), gamma_g2=g2(vk['gamma_g2']), delta_g2=g2(vk['delta_g2']), ic=[g1(x.tobytes()) for x in vk['ic']])
print('pairing check:', ref.verify(pk, [v for k in range(copies) for v in pub[k % distinct]], ref.proof_from_borsh(proof.tobytes())), flush=True)
