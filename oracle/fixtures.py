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


def fast_r1cs(seed, num_gates, num_input, num_aux):
    """Large satisfiable R1CS built directly in CSR/Montgomery form (for 2^16..2^20-gate tests):
    gate k is  (z[i] + ka * z[j]) * (kb * z[l]) = prod * ONE  with the witness chosen first, so rows are
    independent.  A draws from aux variables, B from inputs and aux, C is a multiple of Input(0) = ONE.
    Returns (c_oracle.R1csC, z Montgomery (nv,4) uint64, z_in ints, z_aux ints)."""
    import random
    R = ref.R
    rnd = random.Random(seed)
    nv = num_input + num_aux
    z_in = [1] + [rnd.randrange(R) for _ in range(num_input - 1)]
    z_aux = [rnd.randrange(R) if rnd.random() < 0.5 else rnd.randrange(2) for _ in range(num_aux)]
    zs = z_in + z_aux
    MONT = ref.MONT_R % R
    one_m = MONT
    a_col, a_val, b_col, b_val, c_col, c_val, lens = [], [], [], [], [], [], []
    for _ in range(num_gates):
        i = num_input + rnd.randrange(num_aux)
        j = num_input + rnd.randrange(num_aux)
        l = rnd.randrange(nv)
        ka = rnd.randrange(1, R)
        kb = rnd.randrange(1, R) if rnd.random() < 0.5 else 1
        av = (zs[i] + ka * zs[j]) % R if j != i else (zs[i] * (1 + ka)) % R
        prod = av * (kb * zs[l] % R) % R
        if j != i:
            a_col += [i, j]; a_val += [one_m, ka * MONT % R]; lens.append(2)
        else:
            a_col += [i]; a_val += [(1 + ka) % R * MONT % R]; lens.append(1)
        b_col.append(l); b_val.append(kb * MONT % R)
        c_col.append(0); c_val.append(prod * MONT % R)
    a_ptr = np.zeros(num_gates + 1, np.uint64)
    a_ptr[1:] = np.cumsum(np.array(lens, np.uint64))
    seq = np.arange(num_gates + 1, dtype=np.uint64)
    A = co.Csr(a_ptr, np.array(a_col, np.uint32), co.limbs_arr(a_val))
    B = co.Csr(seq, np.array(b_col, np.uint32), co.limbs_arr(b_val))
    Cm = co.Csr(seq, np.array(c_col, np.uint32), co.limbs_arr(c_val))
    cs = co.R1csC(num_input, num_aux, A, B, Cm)
    z = co.limbs_arr([x * MONT % R for x in zs])
    return cs, z, z_in, z_aux


def tile_r1cs(csr, copies):
    """`copies` independent instances of one constraint system as ONE system (a batch circuit): the constant ONE is
    shared, every copy gets its own public inputs and aux variables.  Variable order: ONE, copy 0's inputs, copy 1's
    inputs, ..., copy 0's aux, copy 1's aux, ...  Returns a c_oracle.R1csC."""
    nin1, naux1 = csr.num_input - 1, csr.num_aux
    nin, naux = 1 + copies * nin1, copies * naux1

    def tile(m):
        per = len(m.col)
        col = np.tile(m.col.astype(np.int64), copies).reshape(copies, per)
        k = np.arange(copies, dtype=np.int64)[:, None]
        is_one, is_in = col == 0, (col > 0) & (col < csr.num_input)
        out = np.where(is_one, 0, np.where(is_in, col + k * nin1, nin + k * naux1 + (col - csr.num_input)))
        ptr = (m.ptr[None, :-1].astype(np.int64) + k * per).reshape(-1)
        return co.Csr(np.append(ptr, copies * per).astype(np.uint64), out.reshape(-1).astype(np.uint32), np.tile(m.val, (copies, 1)))

    return co.R1csC(nin, naux, tile(csr.A), tile(csr.B), tile(csr.C))


def tile_witness(z_ins, z_auxs):
    """witness of tile_r1cs from the per-copy (z_in, z_aux) lists (z_in[0] == 1 in each)"""
    return witness_mont([1] + [v for zi in z_ins for v in zi[1:]], [v for za in z_auxs for v in za])
