"""
ORACLE (test infrastructure, not product code): restatement of the part of fawkes-crypto's circuit DSL that the
reference's own Groth16 test exercises -- BASELINE.json configs[0] / SURVEY.md section 8f row 3:
`tests/bellman_groth16.rs:19-48` (poseidon merkle proof, depth 32, PoseidonParams::new(3, 8, 53)) -- and, in the second
half of the file, of configs[2]: the eddsa-poseidon signature check over JubJubBN256.

It produces the constraint system (a bn254_ref.R1CS) and a satisfying witness in the exact variable / gate order
the reference emits, so that the product path can be run on the reference's real configs[0] workload instead of a
synthetic one.  Only tests/ and bench.py's checker leg import it.

PARITY UNPINNED for the parameter constants: the reference holds no golden vector for PoseidonParams (they are
derived at run time), and the Rust toolchain is absent here.  What IS pinned:
  * keccak-256 and ChaCha20 against their published known answers (tests/test_fawkes_circuit.py);
  * the gate counts the reference publishes (`README.md:46-52`: poseidon(4,8,54) = 255, merkle proof 32 = 7328);
  * native hash == circuit witness (the constraint system is satisfied by the native evaluation).

Restated from (file:line under /root/reference):
  seedbox/src/lib.rs:9-41                     SeedboxChaCha20 = ChaCha20Rng::from_seed(keccak256(salt)); limbs = next_u64
  ff-uint/src/num/mod.rs:286-303              SeedBoxGen<Num<Fp>>: 4 limbs, shave top bits, accept if < modulus
                                              (the sample IS the Montgomery representation)
  fawkes-crypto/src/native/poseidon.rs:24-96  PoseidonParams::new_with_salt, ark/sigma/mix/perm, poseidon
  fawkes-crypto/src/native/poseidon.rs:124-135 poseidon_merkle_proof_root
  fawkes-crypto/src/circuit/poseidon.rs:16-96 circuit ark/sigma/mix/perm, c_poseidon, c_poseidon_merkle_proof_root
  fawkes-crypto/src/circuit/r1cs/lc.rs:44-138 LC (ordered list; zero coefficients survive only via from_const(0))
  fawkes-crypto/src/circuit/r1cs/num.rs:79-81,136-175,200-262  assert_bit, from_const, switch, assert_eq, +,-,*
  fawkes-crypto/src/circuit/r1cs/bool.rs:19-22,68-71  CBool::alloc = alloc + assert_bit
  fawkes-crypto/src/circuit/r1cs/cs.rs:300-330 BuildCS enforce / inputize / alloc
  fawkes-crypto/src/backend/bellman_groth16/{setup.rs, prover.rs:80-90}  Pub alloc+inputize, then Sec alloc, then circuit
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bn254_ref as ref  # noqa: E402

R = ref.R
M64 = (1 << 64) - 1
RINV = pow(1 << 256, -1, R)


# --------------------------------------------------------------------------- keccak-256 (sha3 crate, Keccak256)
_KECCAK_RC = []
_KECCAK_ROT = [[0] * 5 for _ in range(5)]


def _keccak_init():
    lfsr = 1
    for _ in range(24):
        rc = 0
        for j in range(7):
            if lfsr & 1:
                rc ^= 1 << ((1 << j) - 1)
            lfsr = ((lfsr << 1) ^ (0x71 if lfsr & 0x80 else 0)) & 0xff
        _KECCAK_RC.append(rc)
    x, y = 1, 0
    for t in range(24):
        _KECCAK_ROT[x][y] = ((t + 1) * (t + 2) // 2) % 64
        x, y = y, (2 * x + 3 * y) % 5


_keccak_init()


def _rol64(v, n):
    n %= 64
    return ((v << n) | (v >> (64 - n))) & M64 if n else v


def _keccak_f(a):
    for rnd in range(24):
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol64(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = _rol64(a[x][y], _KECCAK_ROT[x][y])
        a = [[b[x][y] ^ ((~b[(x + 1) % 5][y]) & b[(x + 2) % 5][y]) & M64 for y in range(5)] for x in range(5)]
        a[0][0] ^= _KECCAK_RC[rnd]
    return a


def keccak256(data):
    """original Keccak padding (0x01 ... 0x80), rate 136 -- what sha3::Keccak256 computes"""
    rate = 136
    msg = bytearray(data)
    msg.append(0x01)
    while len(msg) % rate:
        msg.append(0)
    msg[-1] |= 0x80
    a = [[0] * 5 for _ in range(5)]
    for off in range(0, len(msg), rate):
        for i in range(rate // 8):
            a[i % 5][i // 5] ^= int.from_bytes(msg[off + 8 * i:off + 8 * i + 8], 'little')
        a = _keccak_f(a)
    out = b''.join(a[i % 5][i // 5].to_bytes(8, 'little') for i in range(4))
    return out


# --------------------------------------------------------------------------- ChaCha20 (rand_chacha 0.3 ChaCha20Rng)
def _rol32(v, n):
    return ((v << n) | (v >> (32 - n))) & 0xffffffff


def chacha20_block(key_words, counter, stream=0):
    """16 output words of one block; state = consts | key | 64-bit counter | 64-bit stream id (djb layout)"""
    s = [0x61707865, 0x3320646e, 0x79622d32, 0x6b206574] + list(key_words) + [
        counter & 0xffffffff, (counter >> 32) & 0xffffffff, stream & 0xffffffff, (stream >> 32) & 0xffffffff]
    w = list(s)

    def qr(a, b, c, d):
        w[a] = (w[a] + w[b]) & 0xffffffff; w[d] = _rol32(w[d] ^ w[a], 16)
        w[c] = (w[c] + w[d]) & 0xffffffff; w[b] = _rol32(w[b] ^ w[c], 12)
        w[a] = (w[a] + w[b]) & 0xffffffff; w[d] = _rol32(w[d] ^ w[a], 8)
        w[c] = (w[c] + w[d]) & 0xffffffff; w[b] = _rol32(w[b] ^ w[c], 7)

    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(w[i] + s[i]) & 0xffffffff for i in range(16)]


class SeedboxChaCha20:
    """seedbox/src/lib.rs:18-41.  Only next_u64 is ever used, so the word index stays even and a u64 never
    straddles a buffer refill: value = word[2k] | word[2k+1] << 32 of the sequential keystream."""

    def __init__(self, salt):
        seed = keccak256(salt)
        self.key = [int.from_bytes(seed[4 * i:4 * i + 4], 'little') for i in range(8)]
        self.counter = 0
        self.buf = []

    def next_u64(self):
        if not self.buf:
            self.buf = chacha20_block(self.key, self.counter)
            self.counter += 1
        lo, hi = self.buf[0], self.buf[1]
        del self.buf[:2]
        return lo | (hi << 32)

    def gen_fr(self):
        """ff-uint/src/num/mod.rs:286-303 with REPR_SHAVE_BITS = 256 - 254 = 2 for BN254 Fr"""
        while True:
            limbs = [self.next_u64() for _ in range(4)]
            limbs[3] &= M64 >> 2
            mont = limbs[0] | (limbs[1] << 64) | (limbs[2] << 128) | (limbs[3] << 192)
            if mont < R:
                return mont * RINV % R


class PoseidonParams:
    """native/poseidon.rs:24-53"""

    def __init__(self, t, f, p, salt=''):
        sb = SeedboxChaCha20(('fawkes_poseidon(t=%d,f=%d,p=%d,salt=%s)' % (t, f, p, salt)).encode())
        self.t, self.f, self.p = t, f, p
        self.c = [[sb.gen_fr() for _ in range(t)] for _ in range(f + p)]
        x = [sb.gen_fr() for _ in range(t)]
        y = [sb.gen_fr() for _ in range(t)]
        self.m = [[pow(x[i] + y[j], -1, R) for j in range(t)] for i in range(t)]


# --------------------------------------------------------------------------- native (witness-side) hash
def _sigma(a):
    a2 = a * a % R
    return a2 * a2 % R * a % R


def poseidon_perm(state, params):
    half_f = params.f >> 1
    for i in range(params.f + params.p):
        state = [(s + c) % R for s, c in zip(state, params.c[i])]
        if i < half_f or i >= half_f + params.p:
            state = [_sigma(s) for s in state]
        else:
            state[0] = _sigma(state[0])
        state = [sum(params.m[r][j] * state[j] for j in range(params.t)) % R for r in range(params.t)]
    return state


def poseidon(inputs, params):
    assert 0 < len(inputs) < params.t
    state = list(inputs) + [0] * (params.t - len(inputs))
    return poseidon_perm(state, params)[0]


def poseidon_merkle_proof_root(leaf, sibling, path, params):
    root = leaf
    for p, s in zip(path, sibling):
        root = poseidon([s, root] if p else [root, s], params)
    return root


# --------------------------------------------------------------------------- circuit DSL (BuildCS + witness values)
ONE_IDX = (0, 0)   # Index::Input(0); keys sort Input(i) < Aux(j) as lc.rs:155-165


class CS:
    def __init__(self):
        self.num_input = 1
        self.num_aux = 0
        self.gates = []          # (a, b, c) each a list of (coeff, ('i'|'a', idx)) in Index order
        self.z_in = [1]
        self.z_aux = []
        self.const_tracker = []  # one bit per Signal::as_const call (num.rs:111-126, cs.rs:326-328): Parameters.2

    def alloc(self, value):
        v = self.num_aux
        self.num_aux += 1
        self.z_aux.append(value % R)
        return CNum(self, {(1, v): 1}, value % R)

    def enforce(self, a, b, c):
        self.gates.append((a.to_vec(), b.to_vec(), c.to_vec()))

    def inputize(self, n):
        v = self.num_input
        self.num_input += 1
        self.z_in.append(n.value)
        self.gates.append((n.to_vec(), [(1, ('i', 0))], [(1, ('i', v))]))

    def const(self, value):
        return CNum(self, {ONE_IDX: value % R}, value % R)      # from_const keeps a zero coefficient

    def r1cs(self):
        return ref.R1CS(self.num_input, self.num_aux, self.gates)

    def satisfied(self):
        def ev(lc):
            return sum(k * (self.z_in[i] if kind == 'i' else self.z_aux[i]) for k, (kind, i) in lc) % R
        return all(ev(a) * ev(b) % R == ev(c) for a, b, c in self.gates)


class CNum:
    __slots__ = ('cs', 'lc', 'value')

    def __init__(self, cs, lc, value):
        self.cs, self.lc, self.value = cs, lc, value

    def to_vec(self):
        return [(self.lc[k], ('i' if k[0] == 0 else 'a', k[1])) for k in sorted(self.lc)]

    def as_const(self):                      # num.rs:111-126 over lc.rs:68-81
        if not self.lc:
            res = 0
        elif len(self.lc) == 1 and ONE_IDX in self.lc:
            res = self.lc[ONE_IDX]
        else:
            res = None
        self.cs.const_tracker.append(res is not None)
        return res

    def _merge(self, other_lc, sign):        # lc.rs:87-117: sum, drop a term that cancels, insert otherwise
        lc = dict(self.lc)
        for k, v in other_lc.items():
            if k in lc:
                t = (lc[k] + sign * v) % R
                if t:
                    lc[k] = t
                else:
                    del lc[k]
            else:
                lc[k] = (sign * v) % R
        return lc

    def add(self, o):
        return CNum(self.cs, self._merge(o.lc, 1), (self.value + o.value) % R)

    def sub(self, o):
        return CNum(self.cs, self._merge(o.lc, -1), (self.value - o.value) % R)

    def add_const(self, k):
        return self.add(self.cs.const(k))

    def scale(self, k):                      # num.rs:227-236
        k %= R
        if k == 0:
            return self.cs.const(0)
        return CNum(self.cs, {i: v * k % R for i, v in self.lc.items()}, self.value * k % R)

    def mul(self, o):                        # num.rs:247-262
        a, b = self.as_const(), o.as_const()
        if a is not None:
            return o.scale(a)
        if b is not None:
            return self.scale(b)
        signal = self.cs.alloc(self.value * o.value % R)
        self.cs.enforce(self, o, signal)
        return signal

    def assert_bit(self):                    # num.rs:79-81
        self.cs.enforce(self, self.add_const(-1), self.cs.const(0))

    def assert_eq(self, o):                  # num.rs:173-175
        self.cs.enforce(self, self.cs.const(1), o)

    def switch(self, bit, if_else):          # num.rs:161-171
        b = bit.as_const()
        if b is not None:
            return self if b == 1 else if_else
        return if_else.add(self.sub(if_else).mul(bit))


def alloc_bool(cs, bit):                     # bool.rs:68-71
    n = cs.alloc(1 if bit else 0)
    n.assert_bit()
    return n


def c_sigma(a):
    a_sq = a.mul(a)
    a_quad = a_sq.mul(a_sq)
    return a_quad.mul(a)


def c_perm(state, params):
    cs = state[0].cs
    half_f = params.f >> 1
    for i in range(params.f + params.p):
        state = [s.add_const(c) for s, c in zip(state, params.c[i])]
        if i < half_f or i >= half_f + params.p:
            state = [c_sigma(s) for s in state]
        else:
            state[0] = c_sigma(state[0])
        new_state = []
        for r in range(params.t):
            acc = cs.const(0)
            for j in range(params.t):
                acc = acc.add(state[j].scale(params.m[r][j]))
            new_state.append(acc)
        state = new_state
    return state


def c_poseidon(inputs, params):
    assert 0 < len(inputs) < params.t
    cs = inputs[0].cs
    state = list(inputs) + [cs.const(0) for _ in range(params.t - len(inputs))]
    return c_perm(state, params)[0]


def c_poseidon_merkle_proof_root(leaf, sibling, path, params):
    root = leaf
    for p, s in zip(path, sibling):
        first = s.switch(p, root)
        second = root.add(s).sub(first)
        root = c_poseidon([first, second], params)
    return root


def poseidon_merkle_circuit(leaf, sibling, path, depth=32, params=None):
    """tests/bellman_groth16.rs:21-25 under backend prove(): public root (alloc + inputize), secret
    (leaf, CMerkleProof{sibling, path}), circuit.  Returns (CS, root)."""
    params = params or PoseidonParams(3, 8, 53)
    assert len(sibling) == depth and len(path) == depth
    root_value = poseidon_merkle_proof_root(leaf, sibling, path, params)
    cs = CS()
    public = cs.alloc(root_value)
    cs.inputize(public)
    c_leaf = cs.alloc(leaf)
    c_sib = [cs.alloc(s) for s in sibling]
    c_path = [alloc_bool(cs, p) for p in path]
    res = c_poseidon_merkle_proof_root(c_leaf, c_sib, c_path, params)
    res.assert_eq(public)
    return cs, root_value


# =========================================================================== BASELINE configs[2]: poseidon eddsa
# Restated from (file:line under /root/reference/fawkes-crypto/src):
#   engines/bn256/mod.rs:28-75      Fs modulus, JubJubBN256::new (edwards_d, montgomery_a/b/u, generator from the seedbox)
#   native/ecc.rs:56-353            Edwards / Montgomery points, subgroup_decompress, from_scalar_raw, scalar multiplication
#   native/eddsaposeidon.rs:42-79   sign / verify (the nonce here is any scalar: a signature's validity does not depend on
#                                   how rho was derived, so the Blake2s step is not restated)
#   circuit/ecc.rs:25-283           CEdwardsPoint / CMontgomeryPoint gadgets, fixed-base (mux3 windows) and variable-base mul
#   circuit/mux.rs:8-32, circuit/bitify.rs:9-113, circuit/eddsaposeidon.rs:17-47, circuit/r1cs/num.rs:40-78 (div, is_zero)
# Pinned by the reference's published gate counts (README.md:46-53): ecmul_const 254 bits = 513, ecmul 254 bits = 2296,
# poseidon eddsa = 3860.  The curve constants themselves are unpinned (no vector in the reference).
FS = 2736030358979909402780800718157159386076813972158567259200215660948447373041     # jubjub scalar field, 251 bits
FR_BITS, FS_BITS = 254, 251


def fr_inv(a):
    return pow(a % R, -1, R)


def fr_sqrt(a):
    """any square root mod R (every use in the reference normalises the sign afterwards), None for a non-residue"""
    a %= R
    if a == 0:
        return 0
    if pow(a, (R - 1) // 2, R) != 1:
        return None
    s, t = 0, R - 1
    while t % 2 == 0:
        s, t = s + 1, t // 2
    z = 5
    while pow(z, (R - 1) // 2, R) == 1:
        z += 1
    c, r, tt, m = pow(z, t, R), pow(a, (t + 1) // 2, R), pow(a, t, R), s
    while tt != 1:
        i, t2 = 0, tt
        while t2 != 1:
            t2, i = t2 * t2 % R, i + 1
        b = pow(c, 1 << (m - i - 1), R)
        r, c = r * b % R, b * b % R
        tt, m = tt * c % R, i
    return r


class JubJubBN256:
    """engines/bn256/mod.rs:48-75; points are affine (x, y) tuples on  -x^2 + y^2 = 1 + d x^2 y^2"""

    def __init__(self):
        self.d = (-168696) * fr_inv(168700) % R
        self.ma = 2 * (1 - self.d) * fr_inv(1 + self.d) % R
        self.mb = (-4) * fr_inv(1 + self.d) % R
        self.mu = 337401
        self.g = self.from_scalar_raw(SeedboxChaCha20(b'edwards_g').gen_fr())

    def add(self, p, q):                      # unified addition (ecc.rs:309-333 in affine form)
        (x1, y1), (x2, y2) = p, q
        t = self.d * x1 * x2 % R * y1 * y2 % R
        return ((x1 * y2 + y1 * x2) * fr_inv(1 + t) % R, (y1 * y2 + x1 * x2) * fr_inv(1 - t) % R)

    def mul(self, p, k):                      # ecc.rs:339-352
        res = (0, 1)
        for b in bin(k)[2:] if k else '':
            res = self.add(res, res)
            if b == '1':
                res = self.add(res, p)
        return res

    def cofactor(self, p):
        return self.mul(p, 8)

    @staticmethod
    def into_montgomery(p):                   # ecc.rs:182-197
        x, y = p
        if x == 0:
            return None if y == 1 else (0, 0)
        mx = (1 + y) * fr_inv(1 - y) % R
        return (mx, mx * fr_inv(x) % R)

    @staticmethod
    def mont_into_edwards(p):                 # ecc.rs:213-224
        x, y = p
        if x == 0:
            return (0, R - 1)
        return (x * fr_inv(y) % R, (x - 1) * fr_inv(x + 1) % R)

    def from_scalar_raw(self, t):             # ecc.rs:103-132
        g = lambda x: (x * x % R * (x + self.ma) + x) * fr_inv(self.mb) % R
        t2g1 = t * t % R * self.mu % R
        x2 = (-1) * fr_inv(self.ma) * (1 + fr_inv(t2g1)) % R
        y = fr_sqrt(g(x2))
        mx = x2
        if y is None:
            mx = x2 * t2g1 % R
            y = fr_sqrt(g(mx))
        if (y * t % R) & 1:
            y = (-y) % R
        return self.cofactor(self.mont_into_edwards((mx, y)))

    def subgroup_decompress(self, x):         # ecc.rs:71-93
        x2 = x * x % R
        y = fr_sqrt((x2 + 1) * fr_inv(1 - self.d * x2) % R)
        if y is None:
            return None
        lx, ly = self.mul((x, y), FS)
        if lx != 0:
            return None
        return (x, y) if ly == 1 else (x, (-y) % R)


def eddsaposeidon_sign(sk, m, rho, pparams, jj):        # eddsaposeidon.rs:42-53 with the nonce given
    r_x = jj.mul(jj.g, rho)[0]
    a_x = jj.mul(jj.g, sk)[0]
    s = (rho + (poseidon([r_x, a_x, m], pparams) % FS) * sk) % FS
    return s, r_x, a_x


def eddsaposeidon_verify(s, r, a, m, pparams, jj):      # eddsaposeidon.rs:55-79
    p_a, p_r = jj.subgroup_decompress(a), jj.subgroup_decompress(r)
    if p_a is None or p_r is None:
        return False
    ha = jj.mul(p_a, poseidon([r, a, m], pparams) % FS)
    return jj.mul(jj.g, s) == jj.add(ha, p_r)


# ---- more of the DSL (num.rs / bool.rs / bitify.rs / mux.rs / ecc.rs)
def _neg(a):
    return CNum(a.cs, {k: (-v) % R for k, v in a.lc.items()}, (-a.value) % R)


def _rsub(k, a):                              # Num - CNum
    return _neg(a).add_const(k)


def c_square(a):
    return a.mul(a)


def c_div_unchecked(a, b):                    # num.rs:40-51
    ca, cb = a.as_const(), b.as_const()
    if cb is not None:
        return a.scale(fr_inv(cb))
    signal = a.cs.alloc(a.value * (fr_inv(b.value) if b.value else 0) % R)
    a.cs.enforce(signal, b, a)
    return signal


def c_assert_const(a, k):                     # num.rs:153-159 (also CBool::assert_const, bool.rs:73-79)
    a.cs.enforce(a, a.cs.const(1), a.cs.const(k))


def c_is_zero(a):                             # num.rs:69-85 -> CBool (a CNum that is 0 or 1)
    c = a.as_const()
    if c is not None:
        return a.cs.const(1 if c == 0 else 0)
    inv = a.cs.alloc(fr_inv(a.value) if a.value else 0)
    res = _neg(inv).mul(a).add_const(1)
    c_assert_const(res.mul(a), 0)
    return res


def c_into_bits_le(signal, limit):            # bitify.rs:9-48
    cs = signal.cs
    c = signal.as_const()
    if c is not None:
        assert c >> limit == 0
        return [cs.const((c >> i) & 1) for i in range(limit)]
    remained, k = signal, 1
    bits = [cs.const(0)] * limit
    for i in range(1, limit):
        k = 2 * k % R
        s = alloc_bool(cs, (signal.value >> i) & 1)
        remained = remained.sub(s.scale(k))
        bits[i] = s
    remained.assert_bit()                     # to_bool
    bits[0] = remained
    return bits


def c_comp_constant(signal, ct):              # bitify.rs:62-104: true if signal > ct
    siglen = len(signal)
    cs = signal[0].cs
    c_false = cs.const(0)
    if ct >> siglen:
        return c_false
    nsteps = (siglen + 1) >> 1
    k, acc = 1, cs.const(0)
    for i in range(nsteps):
        ct_l, ct_u = (ct >> (2 * i)) & 1, (ct >> (2 * i + 1)) & 1
        sig_l = signal[2 * i]
        sig_u = signal[2 * i + 1] if 2 * i + 1 < siglen else c_false
        sig_lu = sig_l.mul(sig_u)
        if (ct_l, ct_u) == (0, 0):
            term = sig_l.add(sig_u).sub(sig_lu)
        elif (ct_l, ct_u) == (1, 0):
            term = sig_l.add(sig_u.scale(2)).sub(sig_lu).add_const(-1)
        elif (ct_l, ct_u) == (0, 1):
            term = sig_lu.add(sig_u).add_const(-1)
        else:
            term = sig_lu.add_const(-1)
        acc = acc.add(term.scale(k))
        k = 2 * k % R
    acc = acc.add_const(k - 1)
    return c_into_bits_le(acc, nsteps + 1)[nsteps]


def c_into_bits_le_strict(signal):            # bitify.rs:106-111
    bits = c_into_bits_le(signal, FR_BITS)
    c_assert_const(c_comp_constant(bits, R - 1), 0)
    return bits


def c_mux3(s, c):                             # mux.rs:8-32
    s10 = s[0].mul(s[1])
    res = []
    for ci in c:
        a210 = s10.scale(ci[7] - ci[6] - ci[5] + ci[4] - ci[3] + ci[2] + ci[1] - ci[0])
        a21 = s[1].scale(ci[6] - ci[4] - ci[2] + ci[0])
        a20 = s[0].scale(ci[5] - ci[4] - ci[1] + ci[0])
        a2 = ci[4] - ci[0]
        a10 = s10.scale(ci[3] - ci[2] - ci[1] + ci[0])
        a1 = s[1].scale(ci[2] - ci[0])
        a0 = s[0].scale(ci[1] - ci[0])
        res.append(a210.add(a21).add(a20).add_const(a2).mul(s[2]).add(a10).add(a1).add(a0).add_const(ci[0]))
    return res


class CEdwards:
    def __init__(self, x, y):
        self.x, self.y = x, y

    @property
    def cs(self):
        return self.x.cs

    def value(self):
        return (self.x.value, self.y.value)

    def as_const(self):                       # derived Signal::as_const: field by field, short-circuit
        x = self.x.as_const()
        if x is None:
            return None
        y = self.y.as_const()
        return None if y is None else (x, y)

    def switch(self, bit, if_else):
        return CEdwards(self.x.switch(bit, if_else.x), self.y.switch(bit, if_else.y))

    def double(self, jj):                     # ecc.rs:25-33
        v = self.x.mul(self.y)
        v2 = c_square(v)
        u = c_square(self.x.add(self.y))
        return CEdwards(c_div_unchecked(v.scale(2), v2.scale(jj.d).add_const(1)),
                        c_div_unchecked(u.sub(v.scale(2)), _rsub(1, v2.scale(jj.d))))

    def mul_by_cofactor(self, jj):
        return self.double(jj).double(jj).double(jj)

    def add(self, p, jj):                     # ecc.rs:39-48
        v1 = self.x.mul(p.y)
        v2 = p.x.mul(self.y)
        v12 = v1.mul(v2)
        u = self.x.add(self.y).mul(p.x.add(p.y))
        return CEdwards(c_div_unchecked(v1.add(v2), v12.scale(jj.d).add_const(1)),
                        c_div_unchecked(u.sub(v1).sub(v2), _rsub(1, v12.scale(jj.d))))

    def assert_in_curve(self, jj, lean=False):            # ecc.rs:50-55
        x2, y2 = c_square(self.x), c_square(self.y)
        if lean:
            # NOT the source at this revision: the curve equation as ONE gate, (d x^2) * y^2 = y^2 - x^2 - 1, i.e. 3 gates for the
            # check.  Only this form reproduces the README's "jubjub oncurve+subgroup check: 19" (3 + 15 + 1, README.md:47); the
            # source composes `Mul` (a new variable + gate, num.rs:247-262) with `assert_eq` (another gate, num.rs:173-175): 4.
            self.cs.enforce(x2.scale(jj.d), y2, y2.sub(x2).add_const(-1))
            return
        x2.scale(jj.d).mul(y2).assert_eq(y2.sub(x2).add_const(-1))

    @staticmethod
    def subgroup_decompress(x, jj, lean=False):           # ecc.rs:69-80
        p = jj.subgroup_decompress(x.value) or jj.g
        pre = jj.mul(p, fr_inv_mod(8, FS))
        preimage = CEdwards(x.cs.alloc(pre[0]), x.cs.alloc(pre[1]))
        preimage.assert_in_curve(jj, lean)
        p8 = preimage.mul_by_cofactor(jj)
        c_assert_const(x.sub(p8.x), 0)
        return p8

    def into_montgomery(self):                # ecc.rs:83-87
        x = c_div_unchecked(self.y.add_const(1), _rsub(1, self.y))
        return CMont(x, c_div_unchecked(x, self.x))

    def mul(self, bits, jj):                  # ecc.rs:90-186
        cs = self.cs
        c_base = self.as_const()
        if c_base is not None:
            if c_base == (0, 1):
                return CEdwards(cs.const(0), cs.const(1))
            all_bits = list(bits) + [cs.const(0)] * ((2 * len(bits)) % 3)
            nwindows = len(all_bits) // 3
            acc, base = (0, R - 1), c_base
            for _ in range(nwindows):
                acc = jj.add(acc, base)
                base = jj.cofactor(base)
            mp = jj.into_montgomery(((-acc[0]) % R, acc[1]))
            cacc = CMont(cs.const(mp[0]), cs.const(mp[1]))
            base = c_base
            for i in range(nwindows):
                xs, ys, q = [], [], base
                for _ in range(8):
                    mx, my = jj.into_montgomery(q)
                    xs.append(mx); ys.append(my)
                    q = jj.add(q, base)
                res = c_mux3(all_bits[3 * i:3 * i + 3], [xs, ys])
                cacc = cacc.add(CMont(res[0], res[1]), jj)
                base = jj.cofactor(base)
            r = cacc.into_edwards()
            return CEdwards(_neg(r.x), _neg(r.y))
        base_is_zero = c_is_zero(self.x)
        dummy = CEdwards(cs.const(jj.g[0]), cs.const(jj.g[1]))
        base_point = dummy.switch(base_is_zero, self).into_montgomery()
        exponents = [base_point]
        for _ in range(1, len(bits)):
            base_point = base_point.double(jj)
            exponents.append(base_point)
        empty = CMont(cs.const(0), cs.const(0))
        acc = empty
        for i in range(len(bits)):
            acc = acc.add(exponents[i], jj).switch(bits[i], acc)
        acc = empty.switch(base_is_zero, acc)
        r = acc.into_edwards()
        return CEdwards(_neg(r.x), _neg(r.y))


def fr_inv_mod(a, mod):
    return pow(a, -1, mod)


class CMont:
    def __init__(self, x, y):
        self.x, self.y = x, y

    def switch(self, bit, if_else):
        return CMont(self.x.switch(bit, if_else.x), self.y.switch(bit, if_else.y))

    def double(self, jj):                     # ecc.rs:245-256
        x2 = c_square(self.x)
        l = c_div_unchecked(x2.scale(3).add(self.x.scale(2 * jj.ma)).add_const(1), self.y.scale(2 * jj.mb))
        b_l2 = c_square(l).scale(jj.mb)
        return CMont(b_l2.add_const(-jj.ma).sub(self.x.scale(2)),
                     l.mul(self.x.scale(3).add_const(jj.ma).sub(b_l2)).sub(self.y))

    def add(self, p, jj):                     # ecc.rs:259-268
        l = c_div_unchecked(p.y.sub(self.y), p.x.sub(self.x))
        b_l2 = c_square(l).scale(jj.mb)
        return CMont(b_l2.add_const(-jj.ma).sub(self.x).sub(p.x),
                     l.mul(self.x.scale(2).add(p.x).add_const(jj.ma).sub(b_l2)).sub(self.y))

    def into_edwards(self):                   # ecc.rs:271-277
        y_is_zero = c_is_zero(self.y)
        return CEdwards(c_div_unchecked(self.x, self.y.add(y_is_zero)),
                        c_div_unchecked(self.x.add_const(-1), self.x.add_const(1)))


def c_eddsaposeidon_verify(s, r, a, m, pparams, jj, strict_h=True, range_check_s=True, lean_in_curve=False, account=None):
    """circuit/eddsaposeidon.rs:17-47 -> CBool.  The defaults ARE the source at this revision (4121 gates).  The three switches name
    variants that are NOT in the source; they exist only to account for the README's 3860 (README.md:53), tests/test_fawkes_circuit.py:
    strict_h=False   -- h decomposed without the comparator of c_into_bits_le_strict (bitify.rs:107-110): -256 gates;
    range_check_s=False -- without `c_comp_constant(&s_bits, -1 in Fs)` (eddsaposeidon.rs:36): -253 gates;
    lean_in_curve=True  -- the 3-gate curve check that the README's own "oncurve+subgroup check: 19" implies: -1 per decompression.
    account: a list that receives (stage, gates so far)."""
    cs = s.cs
    mark = (lambda name: account.append((name, len(cs.gates)))) if account is not None else (lambda name: None)
    mark('start')
    p_a = CEdwards.subgroup_decompress(a, jj, lean_in_curve); mark('subgroup_decompress(a)')
    p_r = CEdwards.subgroup_decompress(r, jj, lean_in_curve); mark('subgroup_decompress(r)')
    h = c_poseidon([r, a, m], pparams); mark('poseidon(r, a, m)')
    h_bits = c_into_bits_le(h, FR_BITS); mark('h into 254 bits')
    if strict_h:
        c_assert_const(c_comp_constant(h_bits, R - 1), 0)          # = c_into_bits_le_strict (bitify.rs:106-111)
    mark('strict: h <= r - 1 comparator')
    ha = p_a.mul(h_bits, jj); mark('h * A (ecmul, 254 bits)')
    s_bits = c_into_bits_le(s, FS_BITS); mark('s into 251 bits')
    if range_check_s:
        c_assert_const(c_comp_constant(s_bits, FS - 1), 0)
    mark('s <= Fs - 1 comparator')
    sb = CEdwards(cs.const(jj.g[0]), cs.const(jj.g[1])).mul(s_bits, jj); mark('s * G (fixed base, 251 bits)')
    ha_plus_r = ha.add(p_r, jj); mark('hA + R')
    res = c_is_zero(ha_plus_r.x.sub(sb.x)); mark('is_zero')
    return res


def eddsa_circuit(sk, m, rho, pparams=None, jj=None):
    """One signature check as a circuit under the backend's prove(): public input m, secrets (s, r, a), the verifier's
    result asserted true.  Returns (CS, (s, r_x, a_x))."""
    pparams = pparams or PoseidonParams(4, 8, 54)
    jj = jj or JubJubBN256()
    s, r_x, a_x = eddsaposeidon_sign(sk, m, rho, pparams, jj)
    cs = CS()
    c_m = cs.alloc(m)
    cs.inputize(c_m)
    c_s, c_r, c_a = cs.alloc(s), cs.alloc(r_x), cs.alloc(a_x)
    ok = c_eddsaposeidon_verify(c_s, c_r, c_a, c_m, pparams, jj)
    c_assert_const(ok, 1)
    return cs, (s, r_x, a_x)


# =========================================================================== BASELINE configs[3]: a rollup-style transaction
# NOT a circuit of the reference (its rollup lives in other repositories; the README only quotes its size): a transaction
# gadget COMPOSED from the reference's own gadgets restated above, so that "rollup-style R1CS (1024-tx shape)" can be
# exercised with real linear combinations instead of gate statistics.  One transaction updates one account leaf:
#   leaf_old = poseidon(a_x, bal_old), leaf_new = poseidon(a_x, bal_new)
#   merkle(leaf_old, sibling, path) == old_root (public);  merkle(leaf_new, sibling, path) == new_root (public)
#   the owner's eddsa-poseidon signature (s, r) under a_x over the message leaf_new verifies.
def rollup_tx_circuit(sk, bal_old, bal_new, sibling, path, rho, depth=32):
    """Returns the CS of one transaction (2 public inputs besides ONE: old_root, new_root)."""
    assert len(sibling) == depth and len(path) == depth
    p3, p4, jj = PoseidonParams(3, 8, 53), PoseidonParams(4, 8, 54), JubJubBN256()
    a_x = jj.mul(jj.g, sk)[0]
    leaf_old, leaf_new = poseidon([a_x, bal_old], p3), poseidon([a_x, bal_new], p3)
    old_root = poseidon_merkle_proof_root(leaf_old, sibling, path, p3)
    new_root = poseidon_merkle_proof_root(leaf_new, sibling, path, p3)
    s, r_x, a_chk = eddsaposeidon_sign(sk, leaf_new, rho, p4, jj)
    assert a_chk == a_x
    cs = CS()
    c_old_root, c_new_root = cs.alloc(old_root), cs.alloc(new_root)
    cs.inputize(c_old_root)
    cs.inputize(c_new_root)
    c_a, c_bo, c_bn = cs.alloc(a_x), cs.alloc(bal_old), cs.alloc(bal_new)
    c_sib = [cs.alloc(v) for v in sibling]
    c_path = [alloc_bool(cs, p) for p in path]
    c_leaf_old, c_leaf_new = c_poseidon([c_a, c_bo], p3), c_poseidon([c_a, c_bn], p3)
    c_poseidon_merkle_proof_root(c_leaf_old, c_sib, c_path, p3).assert_eq(c_old_root)
    c_poseidon_merkle_proof_root(c_leaf_new, c_sib, c_path, p3).assert_eq(c_new_root)
    c_s, c_r = cs.alloc(s), cs.alloc(r_x)
    c_assert_const(c_eddsaposeidon_verify(c_s, c_r, c_a, c_leaf_new, p4, jj), 1)
    return cs
