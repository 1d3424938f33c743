#!/usr/bin/env python3
"""Writes poseidon_merkle_golden.json: the reference's own Groth16 test circuit (tests/bellman_groth16.rs:19-48,
BASELINE configs[0]) built by oracle/fawkes_circuit.py, proved by the C oracle with the fixed toxic waste of
tests/helpers.py and fixed (r, s), and accepted by the independent python pairing verifier at generation time.
The reference holds no vector for this path ("parity unpinned"); this pins oracle <-> HIP agreement and guards
the circuit restatement against drift.  Run from the repo root:  python tests/golden/make_poseidon_merkle_golden.py
"""
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

SEED = 20261003


def main():
    co.build()
    rnd = random.Random(SEED)
    leaf = rnd.randrange(ref.R)
    sib = [rnd.randrange(ref.R) for _ in range(32)]
    path = [rnd.randrange(2) for _ in range(32)]
    cs, root = fc.poseidon_merkle_circuit(leaf, sib, path)
    assert cs.satisfied()
    csr = fx.r1cs_to_csr(cs.r1cs())
    key = co.setup(csr, **TOXIC)
    z = fx.witness_mont(cs.z_in, cs.z_aux)
    a, b, c, aa, bi, ba = co.synthesize(csr, z)
    r, s = 0x5eed0001, 0x5eed0002
    proof = co.prove(key, a, b, c, z, aa, bi, ba, fx.mont_fr(r), fx.mont_fr(s))
    assert ref.verify(fx.key_to_py(key), [root], ref.proof_from_borsh(proof.tobytes())), 'proof does not verify'
    params = fc.PoseidonParams(3, 8, 53)
    out = dict(
        _doc='poseidon merkle proof depth 32, PoseidonParams(3,8,53); instance = random.Random(seed): leaf, 32 siblings, '
             '32 path bits; toxic waste = tests/helpers.py TOXIC; proof = 256-byte fawkes Borsh, pairing-verified',
        seed=SEED, root='%064x' % root, r='%x' % r, s='%x' % s,
        num_gates=len(cs.gates), num_aux=cs.num_aux, num_input=cs.num_input,
        a_aux_density=int(aa.sum()), b_aux_density=int(ba.sum()),
        poseidon_c0='%064x' % params.c[0][0], poseidon_m00='%064x' % params.m[0][0],
        gate_stream_sha256=hashlib.sha256(params_io.encode_gate_stream(r1cs_product(csr))).hexdigest(),
        const_tracker_sha256=hashlib.sha256(bytes(cs.const_tracker)).hexdigest(),
        proof=proof.tobytes().hex(),
    )
    json.dump(out, open(os.path.join(HERE, 'poseidon_merkle_golden.json'), 'w'), indent=1)
    print('written', out['root'])


if __name__ == '__main__':
    main()
