"""CPU checks of the arithmetic identities and bounds the generated device products rely on (tools/gen_mont_mul.py,
fawkes-crypto_amd/csrc/field.hpp): exact big-integer replays of what the kernels compute, no GPU needed.

* lazily reduced form: a Montgomery reduction of a product of two values below 2p stays below 2p without the final subtraction;
* sums of products with ONE reduction (mulsum, the fused Fq2 product, dot4): the bound before the conditional subtraction of q
  and the range after it;
* the dual squaring: every cross product taken once against a doubled limb (d_j / e_j) gives exactly a^2."""
import random

Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583      # Fq
R_ = 21888242871839275222246405745257275088548364400416034343698204186575808495617     # Fr
M = 1 << 256


def redc(t, p):
    """Montgomery reduction WITHOUT the final subtraction: (t + m p) / 2^256"""
    m = (t * (-pow(p, -1, M))) % M
    u = (t + m * p) >> 256
    assert (t + m * p) % M == 0 and (u * M - t) % p == 0
    return u


def edge_values(p, top):
    rnd = random.Random(7)
    vals = [0, 1, p - 1, p, p + 1, top - 1, top - 2, (1 << 255) - 1 if top > (1 << 255) else top - 3]
    vals += [rnd.randrange(top) for _ in range(40)]
    return [v for v in vals if 0 <= v < top]


def test_lazy_product_stays_below_2p():
    for p in (Q, R_):
        assert 4 * p < M
        vs = edge_values(p, 2 * p)
        for a in vs:
            for b in vs[:12]:
                u = redc(a * b, p)
                assert u < 2 * p and (u * M - a * b) % p == 0


def test_sum_of_two_products_one_reduction():
    # lazy operands (q = 2p): below 1.26 q before the conditional subtraction of q, below q after it; c may be q itself (negq(0))
    for p in (Q, R_):
        q = 2 * p
        vs = edge_values(p, q)
        worst = 0
        for a in vs[:14]:
            for c in vs[:14] + [q]:
                u = redc(a * (q - 1) + c * (q - 1), p)
                worst = max(worst, u)
                assert u < 2 * q
                v = u - q if u >= q else u
                assert v < q
        assert worst < 1.26 * q + 1
    # canonical operands (q = p): below 1.38 p, canonical after one subtraction
    for p in (Q, R_):
        u = redc((p - 1) * (p - 1) + p * (p - 1), p)
        assert u < 2 * p and (u - p if u >= p else u) < p


def test_dot4_canonical_bound():
    for p in (Q, R_):
        u = redc(4 * (p - 1) * (p - 1), p)
        assert u < 2 * p and (u - p if u >= p else u) < p
        assert 5 * p * p < M * p          # five terms would still reduce to below 2p; the kernel uses four


def limbs(x):
    return [(x >> (32 * i)) & 0xffffffff for i in range(8)]


def test_dual_squaring_doubled_limbs_are_exact():
    rnd = random.Random(11)
    cases = [0, 1, (1 << 255) - 1, 0x80000000 * sum(1 << (32 * i) for i in range(7)), 2 * Q - 1, Q, 2 * R_ - 1]
    cases += [rnd.randrange(1 << 255) for _ in range(200)]
    for a in cases:
        assert a < 1 << 255
        v = limbs(a)
        e = [(x << 1) & 0xffffffff for x in v]                                   # e_j = a_j << 1
        d = [e[j] | ((v[j - 1] >> 31) if j else 0) for j in range(8)]           # d_j = e_j | (a_(j-1) >> 31)
        total = 0
        for k in range(15):
            for i in range(max(0, k - 7), min(k, 7) + 1):
                j = k - i
                if i > j:
                    continue
                other = v[j] if i == j else (e[j] if j == i + 1 else d[j])       # the generator's choice of operand
                total += v[i] * other << (32 * k)
        assert total == a * a
