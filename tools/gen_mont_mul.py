#!/usr/bin/env python3
"""Generates fawkes-crypto_amd/csrc/mont_mul_gfx950.inc: the device-side Montgomery product for 8 x u32
limbs as a product-scanning (column-wise) schedule with a 96-bit accumulator (lo:64, hi:32).

Every multiply-accumulate is   v_mad_u64_u32 lo, C, x, y, lo   +   v_addc_co_u32 hi, C, 0, hi, C
(the carry-out of the 64-bit add lands in an SGPR pair C and is folded into `hi` by one v_addc).
Measured on MI355X (tools/mulbench): ~126 G mul/s against 77 G mul/s for the compiler-scheduled CIOS loop,
because v_mad_u64_u32 issues at the same rate as v_mul_lo_u32 (about half the rate of v_add_u32) and the CIOS
form spends half of its VALU slots on v_mov / 64-bit adds.

HAZARD RULE (gfx90a+/gfx950): a VALU instruction that reads an SGPR/VCC written by a previous VALU instruction
needs 2 wait states in between (hipcc pads `s_nop 1` for it in its own code), and nothing inside an asm string is
padded by hipcc.  The schedule is therefore software-pipelined: carries rotate through three SGPR pairs and each
v_addc is emitted two instructions behind its v_mad (mad0 mad1 mad2 addc0 mad3 addc1 ...); columns with fewer
than three macs are padded with s_nop.  mul2 (two independent products) interleaves two such chains, which
satisfies the rule without padding and gives every wave two dependency chains.

All macs of one column half are ONE asm statement.  Result (before the final conditional subtraction) is
identical to CIOS: a*b*2^-256 mod p, < 2p.
"""
import os

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'fawkes-crypto_amd', 'csrc', 'mont_mul_gfx950.inc')


def mad_line(acc, carry, kind, xi, yi):
    # 'vs': y is an SGPR constant -> src0
    if kind == 'vs':
        return 'v_mad_u64_u32 %%%d, %%%d, %%%d, %%%d, %%%d' % (acc, carry, yi, xi, acc)
    return 'v_mad_u64_u32 %%%d, %%%d, %%%d, %%%d, %%%d' % (acc, carry, xi, yi, acc)


def addc_line(hi, carry, fresh=False):
    # fresh: `hi` holds nothing yet (start of a column) -> hi = 0 + 0 + carry, so no zeroing move is needed
    if fresh:
        return 'v_addc_co_u32 %%%d, %%%d, 0, 0, %%%d' % (hi, carry, carry)
    return 'v_addc_co_u32 %%%d, %%%d, 0, %%%d, %%%d' % (hi, carry, hi, carry)


def stmt1(pairs, kind, first=False):
    """single chain; operands: %0 lo, %1 hi, %2..%4 carries, then x,y pairs.  first: first statement of a column"""
    n = len(pairs)
    seq = []                      # ('mad', i) / ('addc', i) / ('nop',)
    for i in range(n):
        seq.append(('mad', i))
        if i >= 2:
            seq.append(('addc', i - 2))
    for i in range(max(0, n - 2), n):      # drain: >= 2 instructions between mad i and addc i
        pos_mad = max(k for k, s in enumerate(seq) if s == ('mad', i))
        gap = len(seq) - pos_mad - 1
        while gap < 2:
            seq.append(('nop',))
            gap += 1
        seq.append(('addc', i))
    idx = 5
    opidx, ops, lines = {}, [], []
    for i, (x, y) in enumerate(pairs):
        opidx[i] = (idx, idx + 1)
        ops.append('"v"(%s), "%s"(%s)' % (x, 's' if kind == 'vs' else 'v', y))
        idx += 2
    fresh = first
    for s in seq:
        if s[0] == 'mad':
            xi, yi = opidx[s[1]]
            lines.append(mad_line(0, 2 + s[1] % 3, kind, xi, yi))
        elif s[0] == 'addc':
            lines.append(addc_line(1, 2 + s[1] % 3, fresh))
            fresh = False
        else:
            lines.append('s_nop 0')
    body = '\\n\\t'.join(lines)
    return '        asm("%s" : "+v"(lo), "%s"(hi), "=&s"(c0), "=&s"(c1), "=&s"(c2) : %s);' % (body, '=&v' if first else '+v', ', '.join(ops))


def stmt2(pairs_a, pairs_b, kind, first=False):
    """two chains; operands: %0 lo, %1 hi, %2 lo2, %3 hi2, %4,%5 carries of A, %6,%7 carries of B, then operands"""
    n = len(pairs_a)
    lines, ops = [], []
    idx = 8
    opidx = {}
    for i in range(n):
        (xa, ya), (xb, yb) = pairs_a[i], pairs_b[i]
        if kind == 'vs':       # the constant is shared by both chains
            opidx[i] = (idx, idx + 2, idx + 1, idx + 2)
            ops.append('"v"(%s), "v"(%s), "s"(%s)' % (xa, xb, ya))
            idx += 3
        else:
            opidx[i] = (idx, idx + 1, idx + 2, idx + 3)
            ops.append('"v"(%s), "v"(%s), "v"(%s), "v"(%s)' % (xa, ya, xb, yb))
            idx += 4
    for i in range(n):
        xa, ya, xb, yb = opidx[i]
        lines.append(mad_line(0, 4 + i % 2, kind, xa, ya))
        lines.append(mad_line(2, 6 + i % 2, kind, xb, yb))
        if i >= 1:
            lines.append(addc_line(1, 4 + (i - 1) % 2, first and i == 1))
            lines.append(addc_line(3, 6 + (i - 1) % 2, first and i == 1))
    if n == 1:
        lines.append('s_nop 0')
    lines.append(addc_line(1, 4 + (n - 1) % 2, first and n == 1))
    lines.append(addc_line(3, 6 + (n - 1) % 2, first and n == 1))
    body = '\\n\\t'.join(lines)
    hc = '=&v' if first else '+v'
    return ('        asm("%s" : "+v"(lo), "%s"(hi), "+v"(lo2), "%s"(hi2), "=&s"(a0), "=&s"(a1), "=&s"(b0), "=&s"(b1) : %s);'
            % (body, hc, hc, ', '.join(ops)))


def columns():
    for k in range(16):
        ab = [(i, k - i) for i in range(max(0, k - 7), min(k, 7) + 1)]
        mp = [(i, k - i) for i in range(max(0, k - 7), min(k - 1, 7) + 1) if k - i >= 1]
        yield k, ab, mp


def gen_mul(o):
    o.append('    static __device__ __forceinline__ Fp mul_body_asm(const Fp &a, const Fp &b) {')
    o.append('        uint64_t lo = 0; uint32_t hi; uint64_t c0, c1, c2;')
    o.append('        uint32_t m0, m1, m2, m3, m4, m5, m6, m7;')
    o.append('        Fp r;')
    for k, ab, mp in columns():
        o.append('        // column %d' % k)
        if mp:
            o.append(stmt1([('m%d' % i, 'P::p(%d)' % j) for i, j in mp], 'vs', first=True))
        if ab:
            o.append(stmt1([('a.v[%d]' % i, 'b.v[%d]' % j) for i, j in ab], 'vv', first=not mp))
        if k < 8:
            o.append('        m%d = (uint32_t)lo * P::INV;' % k)
            o.append(stmt1([('m%d' % k, 'P::p(0)')], 'vs'))
        else:
            o.append('        r.v[%d] = (uint32_t)lo;' % (k - 8))
        if mp or ab:      # `hi` is (re)written by the first v_addc of every column that has products
            o.append('        lo = (lo >> 32) | ((uint64_t)hi << 32);')
    o.append('        return fin(r);')
    o.append('    }')


def gen_mul2(o):
    o.append('// Two independent products at once (r1 = a*b, r2 = c*d): the two column accumulators are interleaved')
    o.append('// instruction by instruction so every wave carries two dependency chains.')
    o.append('    static __device__ __forceinline__ void mul2_body_asm(const Fp &a, const Fp &b, const Fp &c, const Fp &d, Fp &r1, Fp &r2) {')
    o.append('        uint64_t lo = 0, lo2 = 0; uint32_t hi, hi2; uint64_t a0, a1, b0, b1;')
    o.append('        uint32_t m0, m1, m2, m3, m4, m5, m6, m7, n0, n1, n2, n3, n4, n5, n6, n7;')
    o.append('        Fp x, y;')
    for k, ab, mp in columns():
        o.append('        // column %d' % k)
        if mp:   # 8 + 3n operands <= 29 for n <= 7
            o.append(stmt2([('m%d' % i, 'P::p(%d)' % j) for i, j in mp], [('n%d' % i, 'P::p(%d)' % j) for i, j in mp], 'vs', first=True))
        for lo_ in range(0, len(ab), 5):     # 8 + 4n <= 28 operands
            chunk = ab[lo_:lo_ + 5]
            o.append(stmt2([('a.v[%d]' % i, 'b.v[%d]' % j) for i, j in chunk], [('c.v[%d]' % i, 'd.v[%d]' % j) for i, j in chunk], 'vv',
                           first=(not mp and lo_ == 0)))
        if k < 8:
            o.append('        m%d = (uint32_t)lo * P::INV; n%d = (uint32_t)lo2 * P::INV;' % (k, k))
            o.append(stmt2([('m%d' % k, 'P::p(0)')], [('n%d' % k, 'P::p(0)')], 'vs'))
        else:
            o.append('        x.v[%d] = (uint32_t)lo; y.v[%d] = (uint32_t)lo2;' % (k - 8, k - 8))
        if mp or ab:
            o.append('        lo = (lo >> 32) | ((uint64_t)hi << 32); lo2 = (lo2 >> 32) | ((uint64_t)hi2 << 32);')
    o.append('        fin2(x, y, r1, r2);')
    o.append('    }')


def gen_sqr2(o):
    """Two squarings at once (r1 = a^2, r2 = c^2): every cross product a_i a_j, i < j, is taken once against a doubled limb --
    36 multiply-accumulates per square instead of 64.  The doubled limb is d_j = (a_j << 1) | (a_(j-1) >> 31) for j > i + 1
    and e_j = a_j << 1 for j = i + 1: the bit that a_i's doubling carries into limb i + 1 belongs to the pairs of a_i itself
    and must not be counted (sum over j > i of the doubled limbs = 2 * (a's limbs above i) exactly)."""
    o.append('// Two squarings (the caller passes the doubled limbs ad / ae, cd / ce): cross products once; see tools/gen_mont_mul.py')
    o.append('    static __device__ __forceinline__ void sqr2_body_asm(const Fp &a, const Fp &ad, const Fp &ae, const Fp &c, const Fp &cd, const Fp &ce, Fp &r1, Fp &r2) {')
    o.append('        uint64_t lo = 0, lo2 = 0; uint32_t hi, hi2; uint64_t a0, a1, b0, b1;')
    o.append('        uint32_t m0, m1, m2, m3, m4, m5, m6, m7, n0, n1, n2, n3, n4, n5, n6, n7;')
    o.append('        Fp x, y;')
    for k, ab, mp in columns():
        o.append('        // column %d' % k)
        if mp:
            o.append(stmt2([('m%d' % i, 'P::p(%d)' % j) for i, j in mp], [('n%d' % i, 'P::p(%d)' % j) for i, j in mp], 'vs', first=True))
        sq = [(i, j) for i, j in ab if i <= j]
        pa = [('a.v[%d]' % i, ('a.v[%d]' if i == j else ('ae.v[%d]' if j == i + 1 else 'ad.v[%d]')) % j) for i, j in sq]
        pb = [('c.v[%d]' % i, ('c.v[%d]' if i == j else ('ce.v[%d]' if j == i + 1 else 'cd.v[%d]')) % j) for i, j in sq]
        for lo_ in range(0, len(pa), 5):
            o.append(stmt2(pa[lo_:lo_ + 5], pb[lo_:lo_ + 5], 'vv', first=(not mp and lo_ == 0)))
        if k < 8:
            o.append('        m%d = (uint32_t)lo * P::INV; n%d = (uint32_t)lo2 * P::INV;' % (k, k))
            o.append(stmt2([('m%d' % k, 'P::p(0)')], [('n%d' % k, 'P::p(0)')], 'vs'))
        else:
            o.append('        x.v[%d] = (uint32_t)lo; y.v[%d] = (uint32_t)lo2;' % (k - 8, k - 8))
        if mp or ab:
            o.append('        lo = (lo >> 32) | ((uint64_t)hi << 32); lo2 = (lo2 >> 32) | ((uint64_t)hi2 << 32);')
    o.append('        fin2(x, y, r1, r2);')
    o.append('    }')


def gen_mulsum(o):
    """a*b + c*d with ONE Montgomery reduction (single chain): every column holds the products of both terms."""
    o.append('// a*b + c*d with one Montgomery reduction: Y3 = R (Q - X3) - Y1 PPP of the mixed addition (c = -Y1); see tools/gen_mont_mul.py')
    o.append('    static __device__ __forceinline__ Fp mulsum_body_asm(const Fp &a, const Fp &b, const Fp &c, const Fp &d) {')
    o.append('        uint64_t lo = 0; uint32_t hi; uint64_t c0, c1, c2;')
    o.append('        uint32_t m0, m1, m2, m3, m4, m5, m6, m7;')
    o.append('        Fp r;')
    for k, ab, mp in columns():
        o.append('        // column %d' % k)
        if mp:
            o.append(stmt1([('m%d' % i, 'P::p(%d)' % j) for i, j in mp], 'vs', first=True))
        if ab:
            o.append(stmt1([('a.v[%d]' % i, 'b.v[%d]' % j) for i, j in ab], 'vv', first=not mp))
            o.append(stmt1([('c.v[%d]' % i, 'd.v[%d]' % j) for i, j in ab], 'vv'))
        if k < 8:
            o.append('        m%d = (uint32_t)lo * P::INV;' % k)
            o.append(stmt1([('m%d' % k, 'P::p(0)')], 'vs'))
        else:
            o.append('        r.v[%d] = (uint32_t)lo;' % (k - 8))
        if mp or ab:
            o.append('        lo = (lo >> 32) | ((uint64_t)hi << 32);')
    o.append('        return red1q(r);      // a sum of two products: below 2.6 q before this, below q after it')
    o.append('    }')


def gen_dot4(o):
    """a0*b0 + a1*b1 + a2*b2 + a3*b3 with ONE Montgomery reduction (single chain; canonical operands: the sum of four products
    is below 4 p^2, its reduction below 1.76 p, one conditional subtraction makes it canonical)."""
    o.append('// a0*b0 + a1*b1 + a2*b2 + a3*b3 with one Montgomery reduction (rows of the constraint-system evaluation); canonical operands only')
    o.append('    static __device__ __forceinline__ Fp dot4_body_asm(const Fp &a0, const Fp &b0, const Fp &a1, const Fp &b1, const Fp &a2, const Fp &b2, const Fp &a3, const Fp &b3) {')
    o.append('        uint64_t lo = 0; uint32_t hi; uint64_t c0, c1, c2;')
    o.append('        uint32_t m0, m1, m2, m3, m4, m5, m6, m7;')
    o.append('        Fp r;')
    for k, ab, mp in columns():
        o.append('        // column %d' % k)
        if mp:
            o.append(stmt1([('m%d' % i, 'P::p(%d)' % j) for i, j in mp], 'vs', first=True))
        if ab:
            for t in range(4):
                o.append(stmt1([('a%d.v[%d]' % (t, i), 'b%d.v[%d]' % (t, j)) for i, j in ab], 'vv', first=(not mp and t == 0)))
        if k < 8:
            o.append('        m%d = (uint32_t)lo * P::INV;' % k)
            o.append(stmt1([('m%d' % k, 'P::p(0)')], 'vs'))
        else:
            o.append('        r.v[%d] = (uint32_t)lo;' % (k - 8))
        if mp or ab:
            o.append('        lo = (lo >> 32) | ((uint64_t)hi << 32);')
    o.append('        return red1q(r);')
    o.append('    }')


def gen_fq2mul(o):
    """Fq2 product by the schoolbook rule with ONE Montgomery reduction per component: r0 = a0*b0 + a1*nb1 (nb1 = -b1),
    r1 = a0*b1 + a1*b0 -- two column accumulators interleaved like mul2, each column holding the products of both terms.
    400 multiply-accumulates like Karatsuba's three products (408), but none of its five additions / subtractions."""
    o.append('// Fq2 product, schoolbook with one reduction per component (the caller passes nb1 = -b1): see tools/gen_mont_mul.py')
    o.append('    static __device__ __forceinline__ void fq2mul_body_asm(const Fp &a0, const Fp &a1, const Fp &b0, const Fp &b1, const Fp &nb1, Fp &r0, Fp &r1) {')
    o.append('        uint64_t lo = 0, lo2 = 0; uint32_t hi, hi2; uint64_t a0_, a1_, b0_, b1_;'.replace('a0_, a1_, b0_, b1_', 'ca0, ca1, cb0, cb1'))
    o.append('        uint32_t m0, m1, m2, m3, m4, m5, m6, m7, n0, n1, n2, n3, n4, n5, n6, n7;')
    o.append('        Fp x, y;')
    for k, ab, mp in columns():
        o.append('        // column %d' % k)
        if mp:
            o.append(stmt2([('m%d' % i, 'P::p(%d)' % j) for i, j in mp], [('n%d' % i, 'P::p(%d)' % j) for i, j in mp], 'vs', first=True).replace('(a0)', '(ca0)').replace('(a1)', '(ca1)').replace('(b0)', '(cb0)').replace('(b1)', '(cb1)'))
        pa = [('a0.v[%d]' % i, 'b0.v[%d]' % j) for i, j in ab] + [('a1.v[%d]' % i, 'nb1.v[%d]' % j) for i, j in ab]
        pb = [('a0.v[%d]' % i, 'b1.v[%d]' % j) for i, j in ab] + [('a1.v[%d]' % i, 'b0.v[%d]' % j) for i, j in ab]
        for lo_ in range(0, len(pa), 5):
            st = stmt2(pa[lo_:lo_ + 5], pb[lo_:lo_ + 5], 'vv', first=(not mp and lo_ == 0))
            o.append(st.replace('"=&s"(a0)', '"=&s"(ca0)').replace('"=&s"(a1)', '"=&s"(ca1)').replace('"=&s"(b0)', '"=&s"(cb0)').replace('"=&s"(b1)', '"=&s"(cb1)'))
        if k < 8:
            o.append('        m%d = (uint32_t)lo * P::INV; n%d = (uint32_t)lo2 * P::INV;' % (k, k))
            o.append(stmt2([('m%d' % k, 'P::p(0)')], [('n%d' % k, 'P::p(0)')], 'vs').replace('(a0)', '(ca0)').replace('(a1)', '(ca1)').replace('(b0)', '(cb0)').replace('(b1)', '(cb1)'))
        else:
            o.append('        x.v[%d] = (uint32_t)lo; y.v[%d] = (uint32_t)lo2;' % (k - 8, k - 8))
        if mp or ab:
            o.append('        lo = (lo >> 32) | ((uint64_t)hi << 32); lo2 = (lo2 >> 32) | ((uint64_t)hi2 << 32);')
    o.append('        red2q(x, y, r0, r1);      // a sum of two products: below 2.6 q before this, below q after it')
    o.append('    }')


def gen_fq2mulsub(o):
    """Fq2: a*b - c*d with ONE Montgomery reduction per component (Y3 = R (Q - X3) - Y1 PPP of the G2 mixed addition):
    r0 = a0*b0 + a1*nb1 + c0*nd0 + c1*d1,  r1 = a0*b1 + a1*b0 + c0*nd1 + c1*nd0   (n. = the negative as a product operand, q - .)
    -- 8 x 64 + 2 x 72 = 656 multiply-accumulates instead of two fq2mul (800).  A sum of four products is below 4 q^2, its
    reduction below 2.02 q (lazy, q = 2p) or 1.76 q (q = p): two conditional subtractions of q."""
    o.append('// Fq2: a*b - c*d, one reduction per component (the caller passes the negatives nb1, nd0, nd1): see tools/gen_mont_mul.py')
    o.append('    static __device__ __forceinline__ void fq2mulsub_body_asm(const Fp &a0, const Fp &a1, const Fp &b0, const Fp &b1, const Fp &nb1, const Fp &c0, const Fp &c1,'
             ' const Fp &d1, const Fp &nd0, const Fp &nd1, Fp &r0, Fp &r1) {')
    o.append('        uint64_t lo = 0, lo2 = 0; uint32_t hi, hi2; uint64_t ca0, ca1, cb0, cb1;')
    o.append('        uint32_t m0, m1, m2, m3, m4, m5, m6, m7, n0, n1, n2, n3, n4, n5, n6, n7;')
    o.append('        Fp x, y;')
    ren = lambda st: st.replace('(a0)', '(ca0)').replace('(a1)', '(ca1)').replace('(b0)', '(cb0)').replace('(b1)', '(cb1)')
    for k, ab, mp in columns():
        o.append('        // column %d' % k)
        if mp:
            o.append(ren(stmt2([('m%d' % i, 'P::p(%d)' % j) for i, j in mp], [('n%d' % i, 'P::p(%d)' % j) for i, j in mp], 'vs', first=True)))
        pa, pb = [], []
        for (xa, ya), (xb, yb) in ((('a0', 'b0'), ('a0', 'b1')), (('a1', 'nb1'), ('a1', 'b0')), (('c0', 'nd0'), ('c0', 'nd1')), (('c1', 'd1'), ('c1', 'nd0'))):
            pa += [('%s.v[%d]' % (xa, i), '%s.v[%d]' % (ya, j)) for i, j in ab]
            pb += [('%s.v[%d]' % (xb, i), '%s.v[%d]' % (yb, j)) for i, j in ab]
        for lo_ in range(0, len(pa), 5):
            st = stmt2(pa[lo_:lo_ + 5], pb[lo_:lo_ + 5], 'vv', first=(not mp and lo_ == 0))
            o.append(st.replace('"=&s"(a0)', '"=&s"(ca0)').replace('"=&s"(a1)', '"=&s"(ca1)').replace('"=&s"(b0)', '"=&s"(cb0)').replace('"=&s"(b1)', '"=&s"(cb1)'))
        if k < 8:
            o.append('        m%d = (uint32_t)lo * P::INV; n%d = (uint32_t)lo2 * P::INV;' % (k, k))
            o.append(ren(stmt2([('m%d' % k, 'P::p(0)')], [('n%d' % k, 'P::p(0)')], 'vs')))
        else:
            o.append('        x.v[%d] = (uint32_t)lo; y.v[%d] = (uint32_t)lo2;' % (k - 8, k - 8))
        if mp or ab:
            o.append('        lo = (lo >> 32) | ((uint64_t)hi << 32); lo2 = (lo2 >> 32) | ((uint64_t)hi2 << 32);')
    o.append('        Fp s, t;')
    o.append('        red2q(x, y, s, t);        // a sum of four products: below 2.02 q before this ...')
    o.append('        red2q(s, t, r0, r1);      // ... below 1.02 q here, below q after it')
    o.append('    }')


def check(text):
    """Static check of the hazard rule on the generated text: inside every asm string, a v_addc that reads carry
    operand %k must sit >= 2 instructions after the last instruction that wrote %k."""
    import re
    bad = n = 0
    for m in re.finditer(r'asm\("(.*?)" :', text):
        ins = m.group(1).split('\\n\\t')
        lastw = {}
        for i, l in enumerate(ins):
            t = l.replace(',', ' ').split()
            if not t:
                continue
            if t[0] == 'v_mad_u64_u32':
                lastw[t[2]] = i
            elif t[0] == 'v_addc_co_u32':
                c = t[5]
                if c in lastw and i - lastw[c] - 1 < 2:
                    bad += 1
                lastw[t[2]] = i
                n += 1
    return n, bad


def main():
    o = ['// GENERATED by tools/gen_mont_mul.py -- do not edit.',
         '// Product-scanning Montgomery multiplication, 8 x u32 limbs, gfx950 inline asm (see the generator for the',
         '// schedule and the SGPR-carry hazard rule it obeys).']
    gen_mul(o)
    gen_mul2(o)
    gen_sqr2(o)
    gen_mulsum(o)
    gen_dot4(o)
    gen_fq2mul(o)
    gen_fq2mulsub(o)
    text = '\n'.join(o) + '\n'
    n, bad = check(text)
    if bad:
        raise SystemExit('hazard rule violated in %d of %d v_addc' % (bad, n))
    open(OUT, 'w').write(text)
    print('wrote %s (%d v_addc checked against the SGPR-carry hazard rule, 0 violations)' % (os.path.normpath(OUT), n))


if __name__ == '__main__':
    main()
