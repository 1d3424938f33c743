#!/usr/bin/env python3
"""Regenerates the golden vectors in this directory from oracle/bn254_ref.py (python big-int oracle).
The reference holds no golden vectors for the prover path ("parity unpinned"), so
these pin the oracle <-> C oracle <-> HIP three-way agreement; the proof vector is additionally
checked against the Groth16 pairing equation when generated.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..', 'oracle'))
import bn254_ref as ref  # noqa: E402

R = ref.R


def hx(x):
    return '%064x' % x


def main():
    rng = ref.Lcg(20261003)
    # ---- NTT (canonical values in and out)
    ntt = []
    for k in (0, 1, 3, 6):
        n = 1 << k
        v = [rng.below(R) for _ in range(n)]
        w = ref.omega_for(n)
        g = ref.FR_GEN
        coset_in = [x * pow(g, i, R) % R for i, x in enumerate(v)]
        icoset = [x * pow(g, -i, R) % R for i, x in enumerate(ref.intt(v, w))]
        ntt.append(dict(log_n=k, input=[hx(x) for x in v], forward=[hx(x) for x in ref.ntt(v, w)],
                        inverse=[hx(x) for x in ref.intt(v, w)],
                        coset_forward=[hx(x) for x in ref.ntt(coset_in, w)],
                        coset_inverse=[hx(x) for x in icoset]))
    json.dump(dict(_doc='canonical (non-Montgomery) hex values; out[k] = sum_j in[j] w^(jk)', cases=ntt),
              open(os.path.join(HERE, 'ntt_golden.json'), 'w'), indent=1)

    # ---- MSM: special scalars 0, 1, r-1, an infinity base, a repeated base and a +-P pair
    n = 12
    ks = [rng.below(R) for _ in range(n)]
    g1b = [ref.G1.mul(ref.G1_GEN, k) for k in ks]
    g1b[3] = None
    g1b[5] = g1b[4]
    g1b[7] = ref.G1.neg(g1b[6])
    sc = [rng.below(R) for _ in range(n)]
    sc[0], sc[1], sc[2] = 0, 1, R - 1
    sc[5] = sc[4]
    sc[7] = sc[6]
    g2b = [ref.G2.mul(ref.G2_GEN, k) for k in ks[:6]]
    g2b[2] = None
    msm = dict(
        _doc='bases: raw Montgomery LE bytes (group.rs:57-66 layout, zeros = infinity); scalars canonical hex; '
             'result raw Montgomery LE affine',
        g1_bases=[ref.g1_raw_le(p).hex() for p in g1b], g1_scalars=[hx(s) for s in sc],
        g1_result=ref.g1_raw_le(ref.G1.msm(g1b, sc)).hex(),
        g2_bases=[ref.g2_raw_le(p).hex() for p in g2b], g2_scalars=[hx(s) for s in sc[:6]],
        g2_result=ref.g2_raw_le(ref.G2.msm(g2b, sc[:6])).hex(),
    )
    json.dump(msm, open(os.path.join(HERE, 'msm_golden.json'), 'w'), indent=1)

    # ---- full proof for a toy R1CS with fixed toxic waste and fixed (r, s)
    cs, z_in, z_aux = ref.random_r1cs(424242, num_gates=11, num_input=3, num_aux=14)
    tw = dict(tau=rng.below(R), alpha=rng.below(R), beta=rng.below(R), gamma=rng.below(R), delta=rng.below(R))
    r, s = rng.below(R), rng.below(R)
    pk = ref.setup(cs, **tw)
    proof = ref.prove(pk, cs, z_in, z_aux, r, s)
    assert ref.verify(pk, z_in[1:], proof), 'golden proof does not verify'
    a, b, c, *_ = ref.synthesize(cs, z_in, z_aux)
    h = ref.quotient_h(a, b, c, pk['m'])
    rows = [[[[hx(cf), kind, idx] for cf, (kind, idx) in lc] for lc in row] for row in cs.rows]
    out = dict(
        _doc='toy Groth16 instance; r,s and toxic waste canonical hex; proof = 256-byte fawkes Borsh '
             '(prover.rs:39-45); verified with the pairing equation at generation time',
        num_input=cs.num_input, num_aux=cs.num_aux, rows=rows,
        z_in=[hx(x) for x in z_in], z_aux=[hx(x) for x in z_aux],
        toxic={k: hx(v) for k, v in tw.items()}, r=hx(r), s=hx(s),
        m=pk['m'], h=[hx(x) for x in h],
        n_a=len(pk['a']), n_b=len(pk['b_g1']),
        proof=ref.proof_borsh(*proof).hex(),
    )
    json.dump(out, open(os.path.join(HERE, 'proof_golden.json'), 'w'), indent=1)
    print('golden vectors written')


if __name__ == '__main__':
    main()
