"""
ORACLE (test infrastructure, not product code): restatement of the part of fawkes-crypto's circuit DSL that the
reference's own Groth16 test exercises -- BASELINE.json configs[0] / SURVEY.md section 8f row 3:
`tests/bellman_groth16.rs:19-48` (poseidon merkle proof, depth 32, PoseidonParams::new(3, 8, 53)).

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
