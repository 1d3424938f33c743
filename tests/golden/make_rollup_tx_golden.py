#!/usr/bin/env python3
"""Writes rollup_tx_golden.json: one rollup-style transaction (oracle/fawkes_circuit.py: rollup_tx_circuit -- a composition of
the restated gadgets, not a circuit of the reference) proved by the C oracle with the toxic waste of tests/helpers.py and
fixed (r, s), accepted by the python pairing verifier at generation time.  Pins oracle <-> HIP agreement on a 19270-gate
system with 942 k matrix terms and guards the composed gadget against drift.  Run: python tests/golden/make_rollup_tx_golden.py"""
import hashlib
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, '..', '..')
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import bn254_ref as ref  # noqa: E402
import c_oracle as co  # noqa: E402
import fawkes_circuit as fc  # noqa: E402
import fixtures as fx  # noqa: E402
from helpers import TOXIC, r1cs_product  # noqa: E402
from fawkes_crypto_amd import params_io  # noqa: E402

SK, BAL_OLD, BAL_NEW, RHO, SEED = 0x70110aa70110aa, 1000000, 999000, 0xabcdabcdabcd, 20261003


def inputs():
    rnd = random.Random(SEED)
    return [rnd.randrange(ref.R) for _ in range(32)], [rnd.randrange(2) for _ in range(32)]


def main():
    co.build()
    sibling, path = inputs()
    cs = fc.rollup_tx_circuit(SK, BAL_OLD, BAL_NEW, sibling, path, RHO)
    assert cs.satisfied()
    csr = fx.r1cs_to_csr(cs.r1cs())
    key = co.setup(csr, **TOXIC)
    z = fx.witness_mont(cs.z_in, cs.z_aux)
    a, b, c, aa, bi, ba = co.synthesize(csr, z)
    r, s_ = 0x70110001, 0x70110002
    proof = co.prove(key, a, b, c, z, aa, bi, ba, fx.mont_fr(r), fx.mont_fr(s_))
    assert ref.verify(fx.key_to_py(key), cs.z_in[1:], ref.proof_from_borsh(proof.tobytes())), 'proof does not verify'
    out = dict(
        _doc='rollup-style transaction: leaf = poseidon(a_x, balance) (t=3), two depth-32 merkle proofs over one sibling path (old / new '
             'root public), eddsa-poseidon signature over the new leaf; sibling / path = random.Random(seed) draws as in the generator; '
             'toxic waste = tests/helpers.py TOXIC; proof = 256-byte fawkes Borsh, pairing-verified',
        sk='%x' % SK, bal_old=BAL_OLD, bal_new=BAL_NEW, rho='%x' % RHO, seed=SEED, r='%x' % r, s='%x' % s_,
        old_root='%064x' % cs.z_in[1], new_root='%064x' % cs.z_in[2],
        num_gates=len(cs.gates), num_aux=cs.num_aux, num_input=cs.num_input,
        a_aux_density=int(aa.sum()), b_aux_density=int(ba.sum()),
        gate_stream_sha256=hashlib.sha256(params_io.encode_gate_stream(r1cs_product(csr))).hexdigest(),
        proof=proof.tobytes().hex(),
    )
    json.dump(out, open(os.path.join(HERE, 'rollup_tx_golden.json'), 'w'), indent=1)
    print('written', out['old_root'])


if __name__ == '__main__':
    main()
