"""
ORACLE (test infrastructure, NOT product code) -- Python big-integer restatement of the
Groth16/BN254 algorithm behind fawkes-crypto's `backend::bellman_groth16::prover::prove`.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

PARITY STATUS: **parity unpinned**.  The arithmetic of the reference path lives in the crates.io
dependencies `fawkes-crypto-bellman_ce 0.3.5`, `fawkes-crypto-pairing_ce 0.18.1`, `ff_ce 0.7.1`
(/root/reference/Cargo.lock:413-436,495-504), which are not vendored under /root/reference, and
there is no Rust toolchain in this image.  The reference holds no golden vector for this path
(its only prover test asserts `verify(..) == true`, fawkes-crypto/tests/bellman_groth16.rs:45-46).
What pins this oracle instead:
  * the Groth16 pairing equation checked by `verify()` below (independent of the prover code);
  * the well-known BN254 constants (generator (1,2), 2*G1, the G2 generator on the twist);
  * the ff-uint known-answer tests for 4-limb Montgomery fields
    (ff-uint/tests/ff-uint_tests.rs:35-156) run against `oracle/groth16_oracle.c`;
  * three-way agreement Python <-> C <-> HIP on committed vectors in tests/golden/.

In-repo facts this file follows (reference file:line):
  * moduli / generator 7: fawkes-crypto/src/engines/bn256/mod.rs:13,23,24
  * Montgomery convention R = 2^256, 4 x u64 little-endian limbs:
    ff-uint_derive/src/lib.rs:229-253,354-366; fawkes-crypto/src/backend/bellman_groth16/mod.rs:105-137
  * point byte layout x||y (G2: x.c0||x.c1||y.c0||y.c1), all-zero = infinity:
    fawkes-crypto/src/backend/bellman_groth16/group.rs:54-80,88-122
  * proof Borsh layout a(G1) b(G2) c(G1), canonical LE field elements:
    fawkes-crypto/src/backend/bellman_groth16/prover.rs:39-60; ff-uint_derive/src/lib.rs:687-702
  * variable / row ordering: fawkes-crypto/src/circuit/r1cs/cs.rs:255-268,
    fawkes-crypto/src/backend/bellman_groth16/mod.rs:61-102
Out-of-repo algorithm (bellman_ce prover/generator/verifier/domain/multiexp): SURVEY.md Appendix A.
"""

Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583  # Fq
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # Fr
FR_S = 28
FR_GEN = 7
FR_ROOT_OF_UNITY = pow(FR_GEN, (R - 1) >> FR_S, R)
MONT_R = 1 << 256

G1_GEN = (1, 2)
G2_GEN = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)


# --------------------------------------------------------------------------- fields
class F1:
    """Fq as python ints."""
    zero = 0
    one = 1

    @staticmethod
    def add(a, b): return (a + b) % Q
    @staticmethod
    def sub(a, b): return (a - b) % Q
    @staticmethod
    def mul(a, b): return (a * b) % Q
    @staticmethod
    def sqr(a): return (a * a) % Q
    @staticmethod
    def neg(a): return (-a) % Q
    @staticmethod
    def inv(a): return pow(a, -1, Q)
    @staticmethod
    def is_zero(a): return a % Q == 0
    @staticmethod
    def muli(a, k): return (a * k) % Q


class F2:
    """Fq2 = Fq[u]/(u^2+1), elements are (c0, c1)."""
    zero = (0, 0)
    one = (1, 0)

    @staticmethod
    def add(a, b): return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)
    @staticmethod
    def sub(a, b): return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)
    @staticmethod
    def mul(a, b):
        return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)
    @staticmethod
    def sqr(a):
        return ((a[0] * a[0] - a[1] * a[1]) % Q, (2 * a[0] * a[1]) % Q)
    @staticmethod
    def neg(a): return ((-a[0]) % Q, (-a[1]) % Q)
    @staticmethod
    def inv(a):
        n = pow((a[0] * a[0] + a[1] * a[1]) % Q, -1, Q)
        return ((a[0] * n) % Q, (-a[1] * n) % Q)
    @staticmethod
    def is_zero(a): return a[0] % Q == 0 and a[1] % Q == 0
    @staticmethod
    def muli(a, k): return ((a[0] * k) % Q, (a[1] * k) % Q)


G1_B = 3
G2_B = F2.mul((3, 0), F2.inv((9, 1)))  # 3/(9+u)


# --------------------------------------------------------------------------- curves
class Curve:
    """Short Weierstrass y^2 = x^3 + b, a = 0. Affine points are (x, y) or None (infinity).
    Jacobian points are (X, Y, Z) with Z == 0 for infinity."""

    def __init__(self, F, b):
        self.F = F
        self.b = b

    def on_curve(self, P):
        if P is None:
            return True
        F = self.F
        x, y = P
        return F.is_zero(F.sub(F.sqr(y), F.add(F.mul(F.sqr(x), x), self.b)))

    def jac_inf(self):
        return (self.F.one, self.F.one, self.F.zero)

    def to_jac(self, P):
        if P is None:
            return self.jac_inf()
        return (P[0], P[1], self.F.one)

    def to_affine(self, J):
        F = self.F
        if F.is_zero(J[2]):
            return None
        zi = F.inv(J[2])
        zi2 = F.sqr(zi)
        return (F.mul(J[0], zi2), F.mul(J[1], F.mul(zi2, zi)))

    def jac_double(self, P):
        F = self.F
        X, Y, Z = P
        if F.is_zero(Z):
            return P
        A = F.sqr(X)
        B = F.sqr(Y)
        C = F.sqr(B)
        D = F.muli(F.sub(F.sub(F.sqr(F.add(X, B)), A), C), 2)
        E = F.muli(A, 3)
        Fv = F.sqr(E)
        X3 = F.sub(Fv, F.muli(D, 2))
        Y3 = F.sub(F.mul(E, F.sub(D, X3)), F.muli(C, 8))
        Z3 = F.muli(F.mul(Y, Z), 2)
        return (X3, Y3, Z3)

    def jac_add(self, P, Qp):
        F = self.F
        if F.is_zero(P[2]):
            return Qp
        if F.is_zero(Qp[2]):
            return P
        X1, Y1, Z1 = P
        X2, Y2, Z2 = Qp
        Z1Z1 = F.sqr(Z1)
        Z2Z2 = F.sqr(Z2)
        U1 = F.mul(X1, Z2Z2)
        U2 = F.mul(X2, Z1Z1)
        S1 = F.mul(Y1, F.mul(Z2, Z2Z2))
        S2 = F.mul(Y2, F.mul(Z1, Z1Z1))
        H = F.sub(U2, U1)
        r = F.sub(S2, S1)
        if F.is_zero(H):
            if F.is_zero(r):
                return self.jac_double(P)
            return self.jac_inf()
        HH = F.sqr(H)
        HHH = F.mul(H, HH)
        V = F.mul(U1, HH)
        X3 = F.sub(F.sub(F.sqr(r), HHH), F.muli(V, 2))
        Y3 = F.sub(F.mul(r, F.sub(V, X3)), F.mul(S1, HHH))
        Z3 = F.mul(F.mul(Z1, Z2), H)
        return (X3, Y3, Z3)

    def add(self, P, Qp):
        return self.to_affine(self.jac_add(self.to_jac(P), self.to_jac(Qp)))

    def neg(self, P):
        if P is None:
            return None
        return (P[0], self.F.neg(P[1]))

    def jac_mul(self, J, k):
        res = self.jac_inf()
        if k == 0:
            return res
        for bit in bin(k)[2:]:
            res = self.jac_double(res)
            if bit == '1':
                res = self.jac_add(res, J)
        return res

    def mul(self, P, k):
        return self.to_affine(self.jac_mul(self.to_jac(P), k))

    def msm(self, bases, scalars, c=8):
        """Pippenger; any correct MSM yields the same point."""
        assert len(bases) == len(scalars)
        acc = self.jac_inf()
        nwin = (256 + c - 1) // c
        for w in reversed(range(nwin)):
            for _ in range(c):
                acc = self.jac_double(acc)
            buckets = [None] * ((1 << c) - 1)
            for P, s in zip(bases, scalars):
                if P is None:
                    continue
                d = (s >> (w * c)) & ((1 << c) - 1)
                if d:
                    Jp = self.to_jac(P)
                    buckets[d - 1] = Jp if buckets[d - 1] is None else self.jac_add(buckets[d - 1], Jp)
            run = self.jac_inf()
            for bkt in reversed(buckets):
                if bkt is not None:
                    run = self.jac_add(run, bkt)
                acc = self.jac_add(acc, run)
        return self.to_affine(acc)


G1 = Curve(F1, G1_B)
G2 = Curve(F2, G2_B)


# --------------------------------------------------------------------------- byte layouts
def to_mont(x, p):
    return (x * MONT_R) % p


def from_mont(x, p):
    return (x * pow(MONT_R, -1, p)) % p


def fe_le(x):
    return int(x).to_bytes(32, 'little')


def fe_from_le(b):
    return int.from_bytes(b, 'little')


def g1_raw_le(P):
    """`into_raw_uncompressed_le` layout used at group.rs:57-66 / :74-77: Montgomery LE x||y;
    infinity is the all-zero buffer (group.rs:55,71-72)."""
    if P is None:
        return bytes(64)
    return fe_le(to_mont(P[0], Q)) + fe_le(to_mont(P[1], Q))


def g1_from_raw_le(b):
    if b == bytes(64):
        return None
    return (from_mont(fe_from_le(b[:32]), Q), from_mont(fe_from_le(b[32:64]), Q))


def g2_raw_le(P):
    """group.rs:97-103 / :114-119: x.c0||x.c1||y.c0||y.c1, Montgomery LE."""
    if P is None:
        return bytes(128)
    (x0, x1), (y0, y1) = P
    return b''.join(fe_le(to_mont(v, Q)) for v in (x0, x1, y0, y1))


def g2_from_raw_le(b):
    if b == bytes(128):
        return None
    v = [from_mont(fe_from_le(b[i * 32:(i + 1) * 32]), Q) for i in range(4)]
    return ((v[0], v[1]), (v[2], v[3]))


def proof_borsh(A, B, C):
    """256-byte Borsh proof: prover.rs:39-45 + group.rs:16-21,33-39; each Num<Fq> serialises as
    its canonical (non-Montgomery) LE integer (ff-uint_derive/src/lib.rs:687-693)."""
    def g1(P):
        return bytes(64) if P is None else fe_le(P[0]) + fe_le(P[1])
    def g2(P):
        if P is None:
            return bytes(128)
        (x0, x1), (y0, y1) = P
        return fe_le(x0) + fe_le(x1) + fe_le(y0) + fe_le(y1)
    return g1(A) + g2(B) + g1(C)


def proof_from_borsh(b):
    assert len(b) == 256
    v = [fe_from_le(b[i * 32:(i + 1) * 32]) for i in range(8)]
    A = None if v[0] == 0 and v[1] == 0 else (v[0], v[1])
    B = None if not any(v[2:6]) else ((v[2], v[3]), (v[4], v[5]))
    C = None if v[6] == 0 and v[7] == 0 else (v[6], v[7])
    return A, B, C


# --------------------------------------------------------------------------- NTT / quotient
def omega_for(m):
    exp = m.bit_length() - 1
    assert 1 << exp == m and exp <= FR_S
    return pow(FR_ROOT_OF_UNITY, 1 << (FR_S - exp), R)


def ntt(vals, omega):
    """out[k] = sum_j in[j] * omega^(j k); natural order in and out (Appendix A.2)."""
    n = len(vals)
    a = list(vals)
    logn = n.bit_length() - 1
    for i in range(n):
        j = int(bin(i)[2:].zfill(logn)[::-1], 2) if logn else 0
        if i < j:
            a[i], a[j] = a[j], a[i]
    length = 2
    while length <= n:
        wl = pow(omega, n // length, R)
        for s in range(0, n, length):
            w = 1
            for k in range(length // 2):
                u = a[s + k]
                v = a[s + k + length // 2] * w % R
                a[s + k] = (u + v) % R
                a[s + k + length // 2] = (u - v) % R
                w = w * wl % R
        length *= 2
    return a


def intt(vals, omega):
    n = len(vals)
    ninv = pow(n, -1, R)
    return [v * ninv % R for v in ntt(vals, pow(omega, -1, R))]


def quotient_h(a, b, c, m):
    """h = (A*B - C)/Z coefficients, bellman's 3 ifft + 3 coset_fft + 1 icoset_fft recipe
    (SURVEY Appendix A.2). a, b, c are the row evaluations (length <= m)."""
    w = omega_for(m)
    pad = lambda v: list(v) + [0] * (m - len(v))
    g = FR_GEN
    polys = []
    for v in (a, b, c):
        co = intt(pad(v), w)
        gp = 1
        sh = []
        for x in co:
            sh.append(x * gp % R)
            gp = gp * g % R
        polys.append(ntt(sh, w))
    zinv = pow((pow(g, m, R) - 1) % R, -1, R)
    t = [((x * y - z) % R) * zinv % R for x, y, z in zip(*polys)]
    co = intt(t, w)
    ginv = pow(g, -1, R)
    gp = 1
    h = []
    for x in co:
        h.append(x * gp % R)
        gp = gp * ginv % R
    return h[:m - 1]


# --------------------------------------------------------------------------- R1CS + Groth16
class R1CS:
    """Rows are (A, B, C) with each LC a list of (coeff:int, ('i'|'a', index)).  Variable order as
    in cs.rs:255-268: Input(0) is the constant ONE."""

    def __init__(self, num_input, num_aux, rows):
        self.num_input = num_input
        self.num_aux = num_aux
        self.rows = rows


def synthesize(r1cs, z_in, z_aux):
    """Appendix A.1: a,b,c row evaluations + the three density bitmaps; n = #gates + num_input."""
    a, b, c = [], [], []
    a_aux = [0] * r1cs.num_aux
    b_in = [0] * r1cs.num_input
    b_aux = [0] * r1cs.num_aux

    def ev(lc, din, daux):
        acc = 0
        for coeff, (kind, idx) in lc:
            if kind == 'i':
                if din is not None:
                    din[idx] = 1
                acc += coeff * z_in[idx]
            else:
                if daux is not None:
                    daux[idx] = 1
                acc += coeff * z_aux[idx]
        return acc % R

    for (A, B, C) in r1cs.rows:
        a.append(ev(A, None, a_aux))
        b.append(ev(B, b_in, b_aux))
        c.append(ev(C, None, None))
    for i in range(r1cs.num_input):
        a.append(z_in[i])  # input_i * 0 = 0
        b.append(0)
        c.append(0)
    return a, b, c, a_aux, b_in, b_aux


def next_pow2(n):
    m = 1
    while m < n:
        m *= 2
    return m


def setup(r1cs, tau, alpha, beta, gamma, delta, g1=G1_GEN, g2=G2_GEN):
    """Appendix A.4 (generate_parameters)."""
    n = len(r1cs.rows) + r1cs.num_input
    m = next_pow2(n)
    w = omega_for(m)
    # Lagrange basis at tau
    tm = pow(tau, m, R)
    zt = (tm - 1) % R
    minv = pow(m, -1, R)
    L = []
    wj = 1
    for j in range(m):
        L.append(zt * wj % R * minv % R * pow((tau - wj) % R, -1, R) % R)
        wj = wj * w % R
    nv = r1cs.num_input + r1cs.num_aux
    At = [0] * nv
    Bt = [0] * nv
    Ct = [0] * nv
    vidx = lambda kind, idx: idx if kind == 'i' else r1cs.num_input + idx
    for row, (A, B, C) in enumerate(r1cs.rows):
        for coeff, (k, i) in A:
            At[vidx(k, i)] = (At[vidx(k, i)] + coeff * L[row]) % R
        for coeff, (k, i) in B:
            Bt[vidx(k, i)] = (Bt[vidx(k, i)] + coeff * L[row]) % R
        for coeff, (k, i) in C:
            Ct[vidx(k, i)] = (Ct[vidx(k, i)] + coeff * L[row]) % R
    for i in range(r1cs.num_input):
        row = len(r1cs.rows) + i
        At[i] = (At[i] + L[row]) % R
    dinv = pow(delta, -1, R)
    ginv = pow(gamma, -1, R)
    h = []
    coeff = zt * dinv % R
    tp = 1
    for i in range(m - 1):
        h.append(G1.mul(g1, tp * coeff % R))
        tp = tp * tau % R
    a_q = [G1.mul(g1, x) for x in At]
    b1_q = [G1.mul(g1, x) for x in Bt]
    b2_q = [G2.mul(g2, x) for x in Bt]
    ic = [G1.mul(g1, (beta * At[k] + alpha * Bt[k] + Ct[k]) % R * ginv % R) for k in range(r1cs.num_input)]
    l = [G1.mul(g1, (beta * At[k] + alpha * Bt[k] + Ct[k]) % R * dinv % R)
         for k in range(r1cs.num_input, nv)]
    return dict(
        m=m, num_input=r1cs.num_input, num_aux=r1cs.num_aux,
        alpha_g1=G1.mul(g1, alpha), beta_g1=G1.mul(g1, beta), beta_g2=G2.mul(g2, beta),
        gamma_g2=G2.mul(g2, gamma), delta_g1=G1.mul(g1, delta), delta_g2=G2.mul(g2, delta),
        ic=ic, h=h, l=l,
        a=[p for p in a_q if p is not None],
        b_g1=[p for p in b1_q if p is not None],
        b_g2=[p for p in b2_q if p is not None],
    )


def prove(params, r1cs, z_in, z_aux, r, s):
    """Appendix A.1-A.5 with explicit (r, s). Returns affine (A, B, C)."""
    a, b, c, a_aux, b_in, b_aux = synthesize(r1cs, z_in, z_aux)
    m = params['m']
    h = quotient_h(a, b, c, m)
    H = G1.msm(params['h'], h)
    Lq = G1.msm(params['l'], z_aux)
    sa = list(z_in) + [z for z, d in zip(z_aux, a_aux) if d]
    Aq = G1.msm(params['a'], sa)
    sb = [z for z, d in zip(z_in, b_in) if d] + [z for z, d in zip(z_aux, b_aux) if d]
    B1 = G1.msm(params['b_g1'], sb)
    B2 = G2.msm(params['b_g2'], sb)
    gA = G1.add(G1.add(params['alpha_g1'], Aq), G1.mul(params['delta_g1'], r))
    gB = G2.add(G2.add(params['beta_g2'], B2), G2.mul(params['delta_g2'], s))
    gC = G1.add(H, Lq)
    gC = G1.add(gC, G1.mul(Aq, s))
    gC = G1.add(gC, G1.mul(B1, r))
    gC = G1.add(gC, G1.mul(params['alpha_g1'], s))
    gC = G1.add(gC, G1.mul(params['beta_g1'], r))
    gC = G1.add(gC, G1.mul(params['delta_g1'], r * s % R))
    return gA, gB, gC


# --------------------------------------------------------------------------- pairing (verify only)
# Fq12 = Fq[w]/(w^12 - 18 w^6 + 82); Fq2's u maps to w^6 - 9.
_F12_MOD = [82, 0, 0, 0, 0, 0, -18, 0, 0, 0, 0, 0]


def f12_mul(a, b):
    t = [0] * 23
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                t[i + j] += x * y
    for k in range(22, 11, -1):
        top = t[k]
        if top:
            t[k] = 0
            t[k - 12] -= 82 * top
            t[k - 6] += 18 * top
    return [x % Q for x in t[:12]]


def f12_one():
    return [1] + [0] * 11


def f12_pow(a, e):
    res = f12_one()
    base = a
    while e:
        if e & 1:
            res = f12_mul(res, base)
        base = f12_mul(base, base)
        e >>= 1
    return res


def f12_inv(a):
    # extended Euclid over Fq[w]
    lm, hm = [1] + [0] * 12, [0] * 13
    low, high = list(a) + [0], [82, 0, 0, 0, 0, 0, (-18) % Q, 0, 0, 0, 0, 0, 1]

    def deg(p):
        d = len(p) - 1
        while d and p[d] % Q == 0:
            d -= 1
        return d

    while deg(low):
        # r = high // low
        dega, degb = deg(high), deg(low)
        temp = list(high)
        r_ = [0] * 13
        binv = pow(low[degb], -1, Q)
        for i in range(dega - degb, -1, -1):
            qcoef = temp[degb + i] * binv % Q
            r_[i] = qcoef
            for c_ in range(degb + 1):
                temp[c_ + i] = (temp[c_ + i] - qcoef * low[c_]) % Q
        nm = list(hm)
        new = list(high)
        for i in range(13):
            for j in range(13 - i):
                nm[i + j] = (nm[i + j] - lm[i] * r_[j]) % Q
                new[i + j] = (new[i + j] - low[i] * r_[j]) % Q
        lm, low, hm, high = nm, new, lm, low
    c0inv = pow(low[0], -1, Q)
    return [x * c0inv % Q for x in lm[:12]]


def _f12_from_fq2(x):
    c = [0] * 12
    c[0] = (x[0] - 9 * x[1]) % Q
    c[6] = x[1] % Q
    return c


_W2 = [0, 0, 1] + [0] * 9
_W3 = [0, 0, 0, 1] + [0] * 8


class _F12ops:
    zero = [0] * 12
    one = f12_one()
    @staticmethod
    def add(a, b): return [(x + y) % Q for x, y in zip(a, b)]
    @staticmethod
    def sub(a, b): return [(x - y) % Q for x, y in zip(a, b)]
    @staticmethod
    def mul(a, b): return f12_mul(a, b)
    @staticmethod
    def sqr(a): return f12_mul(a, a)
    @staticmethod
    def neg(a): return [(-x) % Q for x in a]
    @staticmethod
    def inv(a): return f12_inv(a)
    @staticmethod
    def is_zero(a): return all(x % Q == 0 for x in a)
    @staticmethod
    def muli(a, k): return [(x * k) % Q for x in a]


def _twist(P):
    x, y = P
    return (f12_mul(_f12_from_fq2(x), _W2), f12_mul(_f12_from_fq2(y), _W3))


def _cast_g1(P):
    return ([P[0]] + [0] * 11, [P[1]] + [0] * 11)


def _linefunc(P1, P2, T):
    F = _F12ops
    x1, y1 = P1
    x2, y2 = P2
    xt, yt = T
    if x1 != x2:
        m = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
        return F.sub(F.mul(m, F.sub(xt, x1)), F.sub(yt, y1))
    elif y1 == y2:
        m = F.mul(F.muli(F.sqr(x1), 3), F.inv(F.muli(y1, 2)))
        return F.sub(F.mul(m, F.sub(xt, x1)), F.sub(yt, y1))
    else:
        return F.sub(xt, x1)


def _aff_add12(P1, P2):
    F = _F12ops
    if P1 is None:
        return P2
    if P2 is None:
        return P1
    x1, y1 = P1
    x2, y2 = P2
    if x1 == x2:
        if y1 == y2:
            m = F.mul(F.muli(F.sqr(x1), 3), F.inv(F.muli(y1, 2)))
        else:
            return None
    else:
        m = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    nx = F.sub(F.sub(F.sqr(m), x1), x2)
    ny = F.sub(F.mul(m, F.sub(x1, nx)), y1)
    return (nx, ny)


ATE_LOOP_COUNT = 29793968203157093288
LOG_ATE_LOOP_COUNT = 63


def miller_loop(Qp, P):
    """Qp: affine G2 point (Fq2 coords), P: affine G1 point. Returns un-exponentiated Fq12."""
    if Qp is None or P is None:
        return f12_one()
    Qt = _twist(Qp)
    Pt = _cast_g1(P)
    Rp = Qt
    f = f12_one()
    for i in range(LOG_ATE_LOOP_COUNT, -1, -1):
        f = f12_mul(f12_mul(f, f), _linefunc(Rp, Rp, Pt))
        Rp = _aff_add12(Rp, Rp)
        if ATE_LOOP_COUNT & (1 << i):
            f = f12_mul(f, _linefunc(Rp, Qt, Pt))
            Rp = _aff_add12(Rp, Qt)
    Q1 = (f12_pow(Qt[0], Q), f12_pow(Qt[1], Q))
    nQ2 = (f12_pow(Q1[0], Q), _F12ops.neg(f12_pow(Q1[1], Q)))
    f = f12_mul(f, _linefunc(Rp, Q1, Pt))
    Rp = _aff_add12(Rp, Q1)
    f = f12_mul(f, _linefunc(Rp, nQ2, Pt))
    return f


def final_exp(f):
    return f12_pow(f, (Q ** 12 - 1) // R)


def pairing(Qp, P):
    return final_exp(miller_loop(Qp, P))


def verify(params, public_inputs, proof):
    """Appendix A.5: e(A,B) = e(alpha,beta) e(sum x_i ic_i, gamma) e(C,delta); `public_inputs`
    excludes the leading ONE (prover.rs:84-87, verifier.rs:75-76)."""
    A, B, C = proof
    ic = params['ic']
    if len(public_inputs) + 1 != len(ic):
        return False
    acc = G1.to_jac(ic[0])
    for x, p in zip(public_inputs, ic[1:]):
        acc = G1.jac_add(acc, G1.jac_mul(G1.to_jac(p), x % R))
    acc = G1.to_affine(acc)
    f = miller_loop(B, A)
    f = f12_mul(f, miller_loop(params['beta_g2'], G1.neg(params['alpha_g1'])))
    f = f12_mul(f, miller_loop(params['gamma_g2'], G1.neg(acc)))
    f = f12_mul(f, miller_loop(params['delta_g2'], G1.neg(C)))
    return final_exp(f) == f12_one()


# --------------------------------------------------------------------------- deterministic inputs
class Lcg:
    """Tiny deterministic generator so fixtures can be regenerated anywhere (no numpy needed)."""

    def __init__(self, seed):
        self.s = (seed * 0x9E3779B97F4A7C15 + 0xD1B54A32D192ED03) & ((1 << 64) - 1)

    def u64(self):
        # splitmix64
        self.s = (self.s + 0x9E3779B97F4A7C15) & ((1 << 64) - 1)
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & ((1 << 64) - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & ((1 << 64) - 1)
        return z ^ (z >> 31)

    def below(self, p):
        v = 0
        for _ in range(5):
            v = (v << 64) | self.u64()
        return v % p


def random_r1cs(seed, num_gates, num_input, num_aux, max_terms=3):
    """Satisfiable random sparse R1CS: each gate is  (sum a_k z_k) * (sum b_k z_k) = c * z_new
    style, with z_aux chosen so that every row holds.  Returns (R1CS, z_in, z_aux)."""
    rng = Lcg(seed)
    z_in = [1] + [rng.below(R) for _ in range(num_input - 1)]
    z_aux = [None] * num_aux
    rows = []
    # the first aux values are free witnesses; later ones are defined by gates
    free = max(1, num_aux - num_gates)
    for j in range(min(free, num_aux)):
        z_aux[j] = rng.below(R) if rng.u64() % 4 else rng.u64() % 2  # witness-like: some bits
    known = [('i', i) for i in range(num_input)] + [('a', j) for j in range(min(free, num_aux))]

    def val(v):
        return z_in[v[1]] if v[0] == 'i' else z_aux[v[1]]

    def rand_lc():
        k = 1 + rng.u64() % max_terms
        seen = {}
        for _ in range(k):
            v = known[rng.u64() % len(known)]
            coeff = rng.below(R) if rng.u64() % 2 else 1
            if coeff:
                seen[v] = coeff
        return [(cf, v) for v, cf in seen.items()]

    nxt = min(free, num_aux)
    for g in range(num_gates):
        A = rand_lc()
        B = rand_lc()
        av = sum(cf * val(v) for cf, v in A) % R
        bv = sum(cf * val(v) for cf, v in B) % R
        prod = av * bv % R
        if nxt < num_aux:
            z_aux[nxt] = prod
            C = [(1, ('a', nxt))]
            known.append(('a', nxt))
            nxt += 1
        else:
            # express prod as coeff * known nonzero variable
            v = known[rng.u64() % len(known)]
            while val(v) == 0:
                v = known[rng.u64() % len(known)]
            C = [(prod * pow(val(v), -1, R) % R, v)] if prod else []
        rows.append((A, B, C))
    for j in range(num_aux):
        if z_aux[j] is None:
            z_aux[j] = rng.below(R)
    return R1CS(num_input, num_aux, rows), z_in, z_aux
