#!/usr/bin/env python3
"""Writes eddsa_golden.json: one eddsa-poseidon signature check (BASELINE configs[2]'s unit) built by
oracle/fawkes_circuit.py, proved by the C oracle with the toxic waste of tests/helpers.py and fixed (r, s), accepted by the
python pairing verifier at generation time.  The reference holds no vector for this path ("parity unpinned"); this pins
oracle <-> HIP agreement and guards the circuit restatement against drift.  Run: python tests/golden/make_eddsa_golden.py"""
import hashlib
import json
import os
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

SK, M, RHO = 0x5eed5eed5eed5eed5eed, 0x1234567890abcdef1234567890abcdef, 0xfeedfeedfeedfeed


def main():
    co.build()
    jj = fc.JubJubBN256()
    cs, (s, r_x, a_x) = fc.eddsa_circuit(SK, M, RHO, fc.PoseidonParams(4, 8, 54), jj)
    assert cs.satisfied()
    csr = fx.r1cs_to_csr(cs.r1cs())
    key = co.setup(csr, **TOXIC)
    z = fx.witness_mont(cs.z_in, cs.z_aux)
    a, b, c, aa, bi, ba = co.synthesize(csr, z)
    r, s_ = 0xedd5a001, 0xedd5a002
    proof = co.prove(key, a, b, c, z, aa, bi, ba, fx.mont_fr(r), fx.mont_fr(s_))
    assert ref.verify(fx.key_to_py(key), [M], ref.proof_from_borsh(proof.tobytes())), 'proof does not verify'
    out = dict(
        _doc='eddsa-poseidon verifier circuit, PoseidonParams(4,8,54), JubJubBN256; secret key / message / nonce below; toxic waste = '
             'tests/helpers.py TOXIC; proof = 256-byte fawkes Borsh, pairing-verified',
        sk='%x' % SK, m='%x' % M, rho='%x' % RHO, r='%x' % r, s='%x' % s_,
        signature=dict(s='%064x' % s, r_x='%064x' % r_x, a_x='%064x' % a_x),
        jubjub_g=['%064x' % jj.g[0], '%064x' % jj.g[1]],
        num_gates=len(cs.gates), num_aux=cs.num_aux, num_input=cs.num_input,
        a_aux_density=int(aa.sum()), b_aux_density=int(ba.sum()),
        gate_stream_sha256=hashlib.sha256(params_io.encode_gate_stream(r1cs_product(csr))).hexdigest(),
        const_tracker_sha256=hashlib.sha256(bytes(cs.const_tracker)).hexdigest(),
        proof=proof.tobytes().hex(),
    )
    json.dump(out, open(os.path.join(HERE, 'eddsa_golden.json'), 'w'), indent=1)
    print('written', out['signature']['r_x'])


if __name__ == '__main__':
    main()
