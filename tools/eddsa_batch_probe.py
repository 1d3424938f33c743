#!/usr/bin/env python3
"""COPIES eddsa-poseidon signature checks as ONE constraint system (BASELINE configs[2] at reduced batch size): GPU
setup, witness -> proof timing, pairing check.  Uses the oracle-side circuit builder, so it is a probe, not product code."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np
import fawkes_circuit as fc, fixtures as fx, bn254_ref as ref
from helpers import r1cs_product, TOXIC
import fawkes_crypto_amd as fk
copies = int(os.environ.get('COPIES', '256'))
t0 = time.time()
distinct = int(os.environ.get('DISTINCT', '64'))
pp, jj = fc.PoseidonParams(4, 8, 54), fc.JubJubBN256()
sigs = [fc.eddsa_circuit(1000003 * (k + 1), 777 + k, 555 + 31 * k, pp, jj)[0] for k in range(distinct)]
one = fx.r1cs_to_csr(sigs[0].r1cs())
batch = fx.tile_r1cs(one, copies)
pick = [sigs[k % distinct] for k in range(copies)]
z = fx.tile_witness([c.z_in for c in pick], [c.z_aux for c in pick])
r1cs = r1cs_product(batch)
print('built %d signatures: %d gates, %d variables, %d matrix terms in %.1f s' % (copies, batch.num_gates, batch.num_input + batch.num_aux,
      sum(len(m.col) for m in (batch.A, batch.B, batch.C)), time.time() - t0), flush=True)
ctx = fk.Context(0)
t0 = time.time()
dk, vk = ctx.setup(r1cs, **{k: fx.mont_fr(v) for k, v in TOXIC.items()})
dr = ctx.load_r1cs(r1cs)
print('setup + load %.1f s, domain 2^%d' % (time.time() - t0, dk.counts()['m'].bit_length() - 1), flush=True)
r, s = fx.mont_fr(11), fx.mont_fr(22)
proof = ctx.prove_witness(dk, dr, z, r, s)
t0 = time.time()
for _ in range(5):
    p2 = ctx.prove_witness(dk, dr, z, r, s)
dt = (time.time() - t0) / 5
assert p2.tobytes() == proof.tobytes()
print('%d signatures per proof: %.2f ms per proof (witness upload included), %.0f signature checks proved per second' % (copies, dt * 1e3, copies / dt), flush=True)
g1 = lambda b: ref.g1_from_raw_le(bytes(b)); g2 = lambda b: ref.g2_from_raw_le(bytes(b))
pk = dict(alpha_g1=g1(vk['alpha_g1']), beta_g2=g2(vk['beta_g2']), gamma_g2=g2(vk['gamma_g2']), delta_g2=g2(vk['delta_g2']), ic=[g1(x.tobytes()) for x in vk['ic']])
print('pairing check:', ref.verify(pk, [c.z_in[1] for c in pick], ref.proof_from_borsh(proof.tobytes())), flush=True)
