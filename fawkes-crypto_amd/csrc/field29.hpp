// BN254 Fq on 9 x 29-bit limbs, Montgomery radix R' = 2^261, lazily reduced -- the arithmetic of the G1 bucket accumulation.
//
// Why: the 8 x 32-bit product (field.hpp, mont_mul_gfx950.inc) spends a v_addc on every v_mad_u64_u32 to catch the carry out
// of its 64-bit accumulator, plus hazard padding around the SGPR carries: 330 issue slots per product, 134 G products/s.
// With 29-bit limbs a column of up to 18 products (< 2^58 each) and the carry-in fit the 64-bit accumulator, so a product is
// 162 + 9 multiply-accumulates and a handful of shifts/masks, written in plain C++ and scheduled by the compiler: 155 G
// products/s in isolation (tools/mulbench/limb29.hip, profiles/r01_mulbench_limb29.log).  R' = 2^261 leaves 7 spare bits
// above p, so nothing is reduced below p on the way: a product of a < A p and b < B p comes out below (A B / 169 + 1) p
// (2^261 / p > 169), sums and differences just add their bounds, and the bounds of the XYZZ mixed addition close at x < 8p,
// y < 4p, zz, zzz < 2p (see Xyzz29::add_mixed).  Values enter from / leave to the resident 8 x 32-bit Montgomery layout
// (radix 2^256) by a re-slicing of the bits: x 2^256 -> x 2^261 is a shift by 5, free while the limbs are cut anyway.
//
// Results are the same group elements, hence the same proof bytes: tests/test_gpu_msm.py, test_gpu_precompute.py.
#pragma once
#include "curve.hpp"

namespace fk {

#if defined(__HIP_DEVICE_COMPILE__)

struct Fq29 {
    static constexpr uint32_t M = (1u << 29) - 1;
    uint32_t v[9];

    // ---- constants (limbs computed at compile time from the 8 x 32 modulus)
    static __device__ __forceinline__ constexpr uint32_t word(int i) { return i < 8 ? FqParams::p(i) : 0u; }
    static __device__ __forceinline__ constexpr uint32_t p29(int k) {           // limb k of p
        const int bit = 29 * k, w = bit >> 5, s = bit & 31;
        const uint64_t two = (uint64_t)word(w) | ((uint64_t)word(w + 1) << 32);
        return (uint32_t)(two >> s) & M;
    }
    static __device__ __forceinline__ constexpr uint32_t inv29() {              // -p^-1 mod 2^29
        uint32_t x = 1;
        for (int i = 0; i < 6; i++) x *= 2u - p29(0) * x;                       // Newton: p^-1 mod 2^32
        return (0u - x) & M;
    }
    // k * p as normalized limbs (k <= 32: below 2^259)
    static __device__ __forceinline__ constexpr uint32_t kp(int k, int limb) {
        uint64_t carry = 0; uint32_t out = 0;
        for (int i = 0; i <= limb; i++) { const uint64_t t = (uint64_t)p29(i) * (uint32_t)k + carry; out = (uint32_t)t & M; carry = t >> 29; if (i == 8) out = (uint32_t)t; }
        return out;
    }

    static __device__ __forceinline__ Fq29 zero() { Fq29 r; for (int i = 0; i < 9; i++) r.v[i] = 0; return r; }
    __device__ __forceinline__ bool is_zero_limbs() const { uint32_t o = 0; for (int i = 0; i < 9; i++) o |= v[i]; return o == 0; }
    // value in [0, 2p): is it congruent to 0?
    __device__ __forceinline__ bool is_zero_mod_p() const {
        uint32_t o = 0, q = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) { o |= v[i]; q |= v[i] ^ p29(i); }
        return o == 0 || q == 0;
    }

    // a * b / 2^261 mod p, below (A B / 169 + 1) p for a < A p, b < B p.  Limbs of a and b below 2^29 (top limb included).
    static __device__ __forceinline__ Fq29 mul(const Fq29 &a, const Fq29 &b) {
        uint64_t acc = 0;
        uint32_t m[9];
        Fq29 r;
        constexpr uint32_t INV = inv29();
#pragma unroll
        for (int k = 0; k < 9; k++) {
#pragma unroll
            for (int i = 0; i <= k; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
            for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * p29(k - i);
            m[k] = ((uint32_t)acc * INV) & M;
            acc += (uint64_t)m[k] * p29(0);
            acc >>= 29;
        }
#pragma unroll
        for (int k = 9; k < 17; k++) {
#pragma unroll
            for (int i = k - 8; i < 9; i++) { acc += (uint64_t)a.v[i] * b.v[k - i]; acc += (uint64_t)m[i] * p29(k - i); }
            r.v[k - 9] = (uint32_t)acc & M;
            acc >>= 29;
        }
        r.v[8] = (uint32_t)acc;
        return r;
    }
    // a * a / 2^261: the cross products a_i a_j (i < j) are formed once and doubled -- 45 + 81 multiply-accumulates instead of 162.
    // A column holds at most 4 cross products (< 2^60, doubled < 2^61), one square and 9 reduction products: below 2^63.
    static __device__ __forceinline__ Fq29 sqr(const Fq29 &a) {
        uint64_t acc = 0;
        uint32_t m[9];
        Fq29 r;
        constexpr uint32_t INV = inv29();
#pragma unroll
        for (int k = 0; k < 17; k++) {
            uint64_t cross = 0;
#pragma unroll
            for (int i = (k > 8 ? k - 8 : 0); 2 * i < k; i++) cross += (uint64_t)a.v[i] * a.v[k - i];
            acc += cross << 1;
            if ((k & 1) == 0) acc += (uint64_t)a.v[k >> 1] * a.v[k >> 1];
            if (k < 9) {
#pragma unroll
                for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * p29(k - i);
                m[k] = ((uint32_t)acc * INV) & M;
                acc += (uint64_t)m[k] * p29(0);
            } else {
#pragma unroll
                for (int i = k - 8; i < 9; i++) acc += (uint64_t)m[i] * p29(k - i);
                r.v[k - 9] = (uint32_t)acc & M;
            }
            acc >>= 29;
        }
        r.v[8] = (uint32_t)acc;
        return r;
    }

    // two independent products column by column in one instruction stream: every multiply-accumulate of one chain has one of
    // the other between itself and its successor (v_mad_u64_u32 into the same accumulator is a dependent issue)
    static __device__ __forceinline__ void mul2(const Fq29 &a, const Fq29 &b, const Fq29 &c, const Fq29 &d, Fq29 &r1, Fq29 &r2) {
        uint64_t acc = 0, bcc = 0;
        uint32_t m[9], n[9];
        Fq29 o1, o2;
        constexpr uint32_t INV = inv29();
#pragma unroll
        for (int k = 0; k < 9; k++) {
#pragma unroll
            for (int i = 0; i <= k; i++) { acc += (uint64_t)a.v[i] * b.v[k - i]; bcc += (uint64_t)c.v[i] * d.v[k - i]; }
#pragma unroll
            for (int i = 0; i < k; i++) { acc += (uint64_t)m[i] * p29(k - i); bcc += (uint64_t)n[i] * p29(k - i); }
            m[k] = ((uint32_t)acc * INV) & M; n[k] = ((uint32_t)bcc * INV) & M;
            acc += (uint64_t)m[k] * p29(0); bcc += (uint64_t)n[k] * p29(0);
            acc >>= 29; bcc >>= 29;
        }
#pragma unroll
        for (int k = 9; k < 17; k++) {
#pragma unroll
            for (int i = k - 8; i < 9; i++) {
                acc += (uint64_t)a.v[i] * b.v[k - i]; bcc += (uint64_t)c.v[i] * d.v[k - i];
                acc += (uint64_t)m[i] * p29(k - i); bcc += (uint64_t)n[i] * p29(k - i);
            }
            o1.v[k - 9] = (uint32_t)acc & M; o2.v[k - 9] = (uint32_t)bcc & M;
            acc >>= 29; bcc >>= 29;
        }
        o1.v[8] = (uint32_t)acc; o2.v[8] = (uint32_t)bcc;
        r1 = o1; r2 = o2;
    }

    // a - b + K p with K >= bound(b) / p: non-negative, normalized.  One signed carry chain.
    template <int K>
    static __device__ __forceinline__ Fq29 sub(const Fq29 &a, const Fq29 &b) {
        Fq29 r; int32_t c = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int32_t t = (int32_t)a.v[i] - (int32_t)b.v[i] + (int32_t)kp(K, i) + c;
            if (i < 8) { r.v[i] = (uint32_t)t & M; c = t >> 29; } else r.v[i] = (uint32_t)t;
        }
        return r;
    }
    // a - b - 2 c + K p
    template <int K>
    static __device__ __forceinline__ Fq29 sub_b_2c(const Fq29 &a, const Fq29 &b, const Fq29 &c2) {
        Fq29 r; int32_t c = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int32_t t = (int32_t)a.v[i] - (int32_t)b.v[i] - 2 * (int32_t)c2.v[i] + (int32_t)kp(K, i) + c;
            if (i < 8) { r.v[i] = (uint32_t)t & M; c = t >> 29; } else r.v[i] = (uint32_t)t;
        }
        return r;
    }

    // ---- the resident layout: 8 x 32-bit words holding X = x 2^256 mod p (< p)
    // limbs of 32 X = x 2^261 (+ a multiple of p): bit 29k - 5 of X is bit 0 of limb k
    static __device__ __forceinline__ Fq29 from_mont256(const Fq &X) {
        Fq29 r;
        r.v[0] = (X.v[0] << 5) & M;
#pragma unroll
        for (int k = 1; k < 9; k++) {
            const int bit = 29 * k - 5, w = bit >> 5, s = bit & 31;
            const uint64_t two = (uint64_t)X.v[w] | ((uint64_t)(w + 1 < 8 ? X.v[w + 1] : 0u) << 32);
            r.v[k] = (uint32_t)(two >> s) & M;
        }
        return r;
    }
    // 32 p - (32 Y): the negated y coordinate of a point, same bound
    static __device__ __forceinline__ Fq29 neg32(const Fq29 &y) { return sub<32>(zero(), y); }
    // lazily reduced value (below 169 p / 2, i.e. any value this file produces) -> fully reduced 8 x 32 Montgomery (radix 2^256)
    __device__ __forceinline__ Fq to_mont256() const {
        // multiply by (2^256 mod p) as a plain integer: x 2^261 * 2^256 / 2^261 = x 2^256, below 2p; then one conditional subtraction
        constexpr uint32_t c256[9] = FK_FQ29_R256;
        Fq29 c;
#pragma unroll
        for (int i = 0; i < 9; i++) c.v[i] = c256[i];
        const Fq29 t = mul(*this, c);
        Fq29 d; int32_t cy = 0;                       // d = t - p (signed chain); its sign decides
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int32_t u = (int32_t)t.v[i] - (int32_t)p29(i) + cy;
            if (i < 8) { d.v[i] = (uint32_t)u & M; cy = u >> 29; } else { d.v[i] = (uint32_t)u; cy = u >> 31; }
        }
        const bool below = cy != 0;                   // t < p: keep t
        Fq o;
#pragma unroll
        for (int w = 0; w < 8; w++) {                 // re-slice 9 x 29 -> 8 x 32 (32 w mod 29 <= 21: two limbs always cover a word)
            const int bit = 32 * w, k = bit / 29, s_ = bit % 29;
            const uint32_t l0 = below ? t.v[k] : d.v[k], l1 = below ? t.v[k + 1] : d.v[k + 1];
            o.v[w] = (uint32_t)(((uint64_t)l0 >> s_) | ((uint64_t)l1 << (29 - s_)));
        }
        return o;
    }
    static __device__ __forceinline__ Fq29 one() {    // 1 in the 2^261 radix, fully reduced
        constexpr uint32_t o[9] = FK_FQ29_ONE;
        Fq29 r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.v[i] = o[i];
        return r;
    }
    // resident value -> the same field element below 2p in the 2^261 radix
    static __device__ __forceinline__ Fq29 lift(const Fq &X) { return mul(from_mont256(X), one()); }
};

// XYZZ accumulator on Fq29.  Invariant between additions: x < 8p, y < 4p, zz, zzz < 2p; infinity = all limbs zero in zz.
struct Xyzz29 {
    Fq29 x, y, zz, zzz;

    static __device__ __forceinline__ Xyzz29 inf() { return Xyzz29{Fq29::zero(), Fq29::zero(), Fq29::zero(), Fq29::zero()}; }
    __device__ __forceinline__ bool is_inf() const { return zz.is_zero_limbs(); }

    // acc += (qx, qy) (affine, resident layout), negated if neg.  madd-2008-s, bounds in units of p on the right.
    __device__ __forceinline__ void add_mixed(const G1Affine &q, bool neg) {
        if (q.is_inf()) return;
        Fq29 qx = Fq29::from_mont256(q.x);                                   // < 32
        Fq29 qy = Fq29::from_mont256(q.y);
        if (neg) qy = Fq29::neg32(qy);                                       // < 32
        if (is_inf()) {          // first entry of the bucket: the point itself, coordinates brought below 2p by a product with 1
            x = Fq29::mul(qx, Fq29::one()); y = Fq29::mul(qy, Fq29::one());  // 32 * 1 / 169 + 1 < 2
            zz = Fq29::one(); zzz = Fq29::one();
            return;
        }
#ifndef FK_L29_SEQUENTIAL
        Fq29 u2, s2, pp, rr, ppp, q_, t, yppp;
        Fq29::mul2(qx, zz, qy, zzz, u2, s2);                                 // 32 * 2 / 169 + 1 < 2
        const Fq29 p = Fq29::sub<8>(u2, x), r = Fq29::sub<4>(s2, y);         // < 10, < 6
        Fq29::mul2(p, p, r, r, pp, rr);                                      // < 2, < 2
        if (pp.is_zero_mod_p()) { exceptional(q, neg, rr.is_zero_mod_p()); return; }
        Fq29::mul2(p, pp, x, pp, ppp, q_);                                   // < 2, < 2
        const Fq29 x3 = Fq29::sub_b_2c<6>(rr, ppp, q_);                      // rr - ppp - 2 q_ + 6p  < 8
        Fq29::mul2(r, Fq29::sub<8>(q_, x3), y, ppp, t, yppp);                // 6 * 10 / 169 + 1 < 2, < 2
        y = Fq29::sub<2>(t, yppp);                                           // < 4
        x = x3;
        Fq29::mul2(zz, pp, zzz, ppp, zz, zzz);                               // 2 * 2 / 169 + 1 < 2
#else
        const Fq29 u2 = Fq29::mul(qx, zz), s2 = Fq29::mul(qy, zzz);          // 32 * 2 / 169 + 1 < 2
        const Fq29 p = Fq29::sub<8>(u2, x), r = Fq29::sub<4>(s2, y);         // < 10, < 6
        const Fq29 pp = Fq29::sqr(p), rr = Fq29::sqr(r);                     // < 2, < 2
        if (pp.is_zero_mod_p()) { exceptional(q, neg, rr.is_zero_mod_p()); return; }
        const Fq29 ppp = Fq29::mul(p, pp), q_ = Fq29::mul(x, pp);            // < 2, < 2
        const Fq29 x3 = Fq29::sub_b_2c<6>(rr, ppp, q_);                      // rr - ppp - 2 q_ + 6p  < 8
        const Fq29 t = Fq29::mul(r, Fq29::sub<8>(q_, x3));                   // 6 * 10 / 169 + 1 < 2
        const Fq29 yppp = Fq29::mul(y, ppp);                                 // < 2
        y = Fq29::sub<2>(t, yppp);                                           // < 4
        x = x3;
        zz = Fq29::mul(zz, pp); zzz = Fq29::mul(zzz, ppp);                   // 2 * 2 / 169 + 1 < 2
#endif
    }

    // P + (+-)P or P - P: through the resident-layout formulas (cold, out-of-line multiply)
    __device__ __noinline__ void exceptional(const G1Affine &q, bool neg, bool same) {
        if (!same) { *this = inf(); return; }
        Affine<FqC> qc; __builtin_memcpy(&qc, &q, sizeof q);
        if (neg) qc.y = FqC::neg(qc.y);
        const Xyzz<FqC> d = Xyzz<FqC>::dbl_affine(qc);
        Fq t;
        __builtin_memcpy(&t, &d.x, 32); x = Fq29::lift(t);
        __builtin_memcpy(&t, &d.y, 32); y = Fq29::lift(t);
        __builtin_memcpy(&t, &d.zz, 32); zz = Fq29::lift(t);
        __builtin_memcpy(&t, &d.zzz, 32); zzz = Fq29::lift(t);
    }

    __device__ __forceinline__ G1Xyzz to_resident() const {
        if (is_inf()) return G1Xyzz::inf();
        return G1Xyzz{x.to_mont256(), y.to_mont256(), zz.to_mont256(), zzz.to_mont256()};
    }
};

#endif  // __HIP_DEVICE_COMPILE__

}  // namespace fk
