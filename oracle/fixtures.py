"""
ORACLE-side helpers (test infrastructure): conversions between the python big-int R1CS of
bn254_ref.py and the CSR/Montgomery form that both oracle/groth16_oracle.c and the product C-ABI
(include/fawkes_hip.h) take.  No product code imports this.
"""
import numpy as np

import bn254_ref as ref
import c_oracle as co


def r1cs_to_csr(cs):
    """bn254_ref.R1CS -> c_oracle.R1csC (variables: Input(i) -> i, Aux(j) -> num_input + j)."""
    mats = []
    for side in range(3):
        ptr = [0]
        col = []
        val = []
        for row in cs.rows:
            for coeff, (kind, idx) in row[side]:
                col.append(idx if kind == 'i' else cs.num_input + idx)
                val.append(ref.to_mont(coeff % ref.R, ref.R))
            ptr.append(len(col))
        v = co.limbs_arr(val) if val else np.zeros((0, 4), np.uint64)
        mats.append(co.Csr(np.array(ptr, np.uint64), np.array(col, np.uint32), v))
    return co.R1csC(cs.num_input, cs.num_aux, *mats)


def witness_mont(z_in, z_aux):
    return co.limbs_arr([ref.to_mont(x % ref.R, ref.R) for x in list(z_in) + list(z_aux)])


def key_to_py(key):
    """c_oracle.Key -> dict in the shape bn254_ref.setup returns (for bn254_ref.verify)."""
    g1 = lambda b: ref.g1_from_raw_le(bytes(b))
    g2 = lambda b: ref.g2_from_raw_le(bytes(b))
    return dict(
        m=key.m, num_input=key.num_input, num_aux=key.num_aux,
        alpha_g1=g1(key.alpha_g1), beta_g1=g1(key.beta_g1), beta_g2=g2(key.beta_g2),
        gamma_g2=g2(key.gamma_g2), delta_g1=g1(key.delta_g1), delta_g2=g2(key.delta_g2),
        ic=[g1(r.tobytes()) for r in key.ic],
    )


def mont_fr(x):
    return co.limbs(ref.to_mont(x % ref.R, ref.R))
