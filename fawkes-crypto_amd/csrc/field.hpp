// BN254 prime fields for gfx950: 8 x u32 little-endian limbs, Montgomery form with R = 2^256.
//
// The limb/Montgomery convention is the one fawkes-crypto hands across the boundary: a `Num<Fr>` is
// `#[repr(transparent)]` over 4 x u64 LE Montgomery limbs (ff-uint/src/num/mod.rs:21-23,
// ff-uint/src/uint/mod.rs:21-26; R/R2/INV rules ff-uint_derive/src/lib.rs:237-253,354-366), and
// bellman receives the same raw limbs (backend/bellman_groth16/mod.rs:105-120).  8 x u32 LE is the
// same byte image, so witness buffers are read as-is.
//
// No MFMA here: this is 256-bit integer modular arithmetic.  On the device the product is a
// product-scanning Montgomery multiplication (mont_mul_gfx950.inc, generated): 128 v_mad_u64_u32, each
// followed by one v_addc that folds the carry-out into a 96-bit column accumulator.  The host pass (and
// the reference for the device code) is the plain CIOS loop `mul_body`.  Every value that leaves a
// function is fully reduced to [0, p) so equality tests are plain limb compares.
#pragma once
#include <stdint.h>
#include "bn254_consts.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#if defined(__HIP_DEVICE_COMPILE__)
#define FK_HD __host__ __device__ __forceinline__   // device pass: hot loops want the CIOS body inline
#else
#define FK_HD __host__ __device__ inline            // host pass: let the compiler decide (build time)
#endif
#else
#define FK_HD inline
#endif

namespace fk {

// p: the modulus of the Montgomery reduction.  q: the modulus additions and subtractions reduce by -- p for the canonical
// form [0, p); 2p for the LAZY form [0, 2p), whose product skips the final conditional subtraction (4p < 2^256, so inputs
// below 2p give a result below 2p) -- ~25 of the product's ~330 instructions.  Lazy values are congruent, not equal: only
// accumulators that are canonicalised before they leave a kernel use them (msm.hip).
struct FqParams {
    static constexpr bool LAZY = false;
    static constexpr FK_HD uint32_t q(int i) { constexpr uint32_t t[8] = FK_FQ_P; return t[i]; }
    static constexpr FK_HD uint32_t p(int i) { constexpr uint32_t t[8] = FK_FQ_P; return t[i]; }
    static constexpr FK_HD uint32_t one(int i) { constexpr uint32_t t[8] = FK_FQ_R; return t[i]; }
    static constexpr FK_HD uint32_t r2(int i) { constexpr uint32_t t[8] = FK_FQ_R2; return t[i]; }
    static constexpr uint32_t INV = FK_FQ_INV;
};
struct FqLazyParams : FqParams {
    using Base = FqParams;
    static constexpr bool LAZY = true;
    static constexpr FK_HD uint32_t q(int i) { constexpr uint32_t t[8] = FK_FQ_2P; return t[i]; }
};
struct FrParams {
    static constexpr bool LAZY = false;
    static constexpr FK_HD uint32_t q(int i) { constexpr uint32_t t[8] = FK_FR_P; return t[i]; }
    static constexpr FK_HD uint32_t p(int i) { constexpr uint32_t t[8] = FK_FR_P; return t[i]; }
    static constexpr FK_HD uint32_t one(int i) { constexpr uint32_t t[8] = FK_FR_R; return t[i]; }
    static constexpr FK_HD uint32_t r2(int i) { constexpr uint32_t t[8] = FK_FR_R2; return t[i]; }
    static constexpr uint32_t INV = FK_FR_INV;
};

// INL = true : the CIOS body is inlined at every use (hot kernels).
// INL = false: on the device the product is an out-of-line call with by-value register arguments --
//              same layout, same results, ~20x less code per point formula (cold kernels: reductions,
//              generators).  The two instantiations are layout-identical and may be reinterpret_cast.
template <class P, bool INL = true>
struct Fp;
#if defined(__HIP_DEVICE_COMPILE__)
template <class P>
__device__ __noinline__ Fp<P, false> mont_mul_call(Fp<P, false> a, Fp<P, false> b);
#endif

template <class P, bool INL>
struct alignas(16) Fp {
    uint32_t v[8];

    static FK_HD Fp zero() { Fp r; for (int i = 0; i < 8; i++) r.v[i] = 0; return r; }
    static FK_HD Fp one() { Fp r; for (int i = 0; i < 8; i++) r.v[i] = P::one(i); return r; }
    static FK_HD Fp r2() { Fp r; for (int i = 0; i < 8; i++) r.v[i] = P::r2(i); return r; }

    FK_HD bool is_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= v[i];
        if constexpr (P::LAZY) {        // 0 or p
            uint32_t e = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) e |= v[i] ^ P::p(i);
            return o == 0 || e == 0;
        }
        return o == 0;
    }
    friend FK_HD bool operator==(const Fp &a, const Fp &b) {
        if constexpr (P::LAZY) return sub(a, b).is_zero();
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
        return o == 0;
    }
    friend FK_HD bool operator!=(const Fp &a, const Fp &b) { return !(a == b); }

    // r = a - p if a >= p (a < 2p)
    static FK_HD Fp reduce_once(const Fp &a) {
        Fp d; uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t t = (uint64_t)a.v[i] - P::p(i) - br;
            d.v[i] = (uint32_t)t; br = (uint32_t)(t >> 63);
        }
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = br ? a.v[i] : d.v[i];
        return r;
    }
    // r = a - q if a >= q (a < 2q): the tail of an addition
    static FK_HD Fp reduce_q(const Fp &a) {
        Fp d; uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t t = (uint64_t)a.v[i] - P::q(i) - br;
            d.v[i] = (uint32_t)t; br = (uint32_t)(t >> 63);
        }
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = br ? a.v[i] : d.v[i];
        return r;
    }
    // the tail of a product: the conditional subtraction of p, or nothing in the lazy form
    static FK_HD Fp fin(const Fp &a) { if constexpr (P::LAZY) return a; else return reduce_once(a); }

    // Addition / subtraction.  Device: generated carry chains (addsub_gfx950.inc, ~30 instructions instead of the ~90 hipcc
    // makes of the C loops below, which remain the host code and the reference).  The *2 forms run two independent
    // operations as four interleaved chains -- use them wherever two operations are independent.
    static FK_HD Fp add_c(const Fp &a, const Fp &b) {
        Fp s; uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t t = (uint64_t)a.v[i] + b.v[i] + c;
            s.v[i] = (uint32_t)t; c = (uint32_t)(t >> 32);
        }
        // p < 2^254 so a + b < 2^255 (lazy: < 2^256): no carry out of limb 7
        return reduce_q(s);
    }
    static FK_HD Fp sub_c(const Fp &a, const Fp &b) {
        Fp d; uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t t = (uint64_t)a.v[i] - b.v[i] - br;
            d.v[i] = (uint32_t)t; br = (uint32_t)(t >> 63);
        }
        uint32_t mask = 0u - br, c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t t = (uint64_t)d.v[i] + (P::q(i) & mask) + c;
            d.v[i] = (uint32_t)t; c = (uint32_t)(t >> 32);
        }
        return d;
    }
#if defined(__HIP_DEVICE_COMPILE__)
#include "addsub_gfx950.inc"
    static FK_HD Fp add(const Fp &a, const Fp &b) { return as1_A(a, b); }
    static FK_HD Fp sub(const Fp &a, const Fp &b) { return as1_S(a, b); }
    static FK_HD void add2(const Fp &a, const Fp &b, const Fp &c, const Fp &d, Fp &r1, Fp &r2) { as2_AA(a, b, c, d, r1, r2); }
    static FK_HD void sub2(const Fp &a, const Fp &b, const Fp &c, const Fp &d, Fp &r1, Fp &r2) { as2_SS(a, b, c, d, r1, r2); }
    static FK_HD void addsub2(const Fp &a, const Fp &b, const Fp &c, const Fp &d, Fp &r1, Fp &r2) { as2_AS(a, b, c, d, r1, r2); }   // r1 = a + b, r2 = c - d
#else
    static FK_HD Fp add(const Fp &a, const Fp &b) { return add_c(a, b); }
    static FK_HD Fp sub(const Fp &a, const Fp &b) { return sub_c(a, b); }
    static FK_HD void add2(const Fp &a, const Fp &b, const Fp &c, const Fp &d, Fp &r1, Fp &r2) { const Fp t = add_c(a, b); r2 = add_c(c, d); r1 = t; }
    static FK_HD void sub2(const Fp &a, const Fp &b, const Fp &c, const Fp &d, Fp &r1, Fp &r2) { const Fp t = sub_c(a, b); r2 = sub_c(c, d); r1 = t; }
    static FK_HD void addsub2(const Fp &a, const Fp &b, const Fp &c, const Fp &d, Fp &r1, Fp &r2) { const Fp t = add_c(a, b); r2 = sub_c(c, d); r1 = t; }
#endif
    static FK_HD Fp dbl(const Fp &a) { return add(a, a); }

    static FK_HD Fp neg(const Fp &a) { return sub(zero(), a); }

    static FK_HD Fp mul(const Fp &a, const Fp &b) {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (!INL) return mont_mul_call<P>(a, b);
        else return mul_body_asm(a, b);
#else
        return mul_body(a, b);
#endif
    }

#if defined(__HIP_DEVICE_COMPILE__)
    static FK_HD void fin2(const Fp &x, const Fp &y, Fp &r1, Fp &r2) { if constexpr (P::LAZY) { r1 = x; r2 = y; } else red2(x, y, r1, r2); }
#include "mont_mul_gfx950.inc"
#endif

    // r1 = a*b, r2 = c*d.  On the device (inlined flavour) the two products run as two interleaved accumulator
    // chains in one instruction stream -- twice the ILP per wave, same instruction count.
    static FK_HD void mul2(const Fp &a, const Fp &b, const Fp &c, const Fp &d, Fp &r1, Fp &r2) {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (INL) { mul2_body_asm(a, b, c, d, r1, r2); return; }
#endif
        r1 = mul(a, b); r2 = mul(c, d);
    }

    // q - b (b <= q): the negative of b as a product operand (Fq2T::mul) -- not a reduced value when b is 0
    static FK_HD Fp negq(const Fp &b) {
        Fp r; uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { const uint64_t t = (uint64_t)P::q(i) - b.v[i] - br; r.v[i] = (uint32_t)t; br = (uint32_t)(t >> 63); }
        return r;
    }
    // (a0 + a1 u)(b0 + b1 u), u^2 = -1, with ONE Montgomery reduction per component (device, inlined flavour): r0 = a0 b0 - a1 b1,
    // r1 = a0 b1 + a1 b0.  The sum of two products is below 2 q^2, its reduction below 1.26 q (lazy: q = 2p) or 1.38 q (q = p),
    // one conditional subtraction of q brings it back under q.
    static constexpr bool HAS_FQ2MUL =
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FK_FQ2_KARATSUBA)
        INL;
#else
        false;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    static __device__ __forceinline__ void fq2mul(const Fp &a0, const Fp &a1, const Fp &b0, const Fp &b1, Fp &r0, Fp &r1) {
        fq2mul_body_asm(a0, a1, b0, b1, negq(b1), r0, r1);
    }
    // (a0 + a1 u)(b0 + b1 u) - (c0 + c1 u)(d0 + d1 u), one reduction per component: 656 multiply-accumulates instead of 800
    static __device__ __forceinline__ void fq2mulsub(const Fp &a0, const Fp &a1, const Fp &b0, const Fp &b1, const Fp &c0, const Fp &c1, const Fp &d0, const Fp &d1,
                                                     Fp &r0, Fp &r1) {
        fq2mulsub_body_asm(a0, a1, b0, b1, negq(b1), c0, c1, d1, negq(d0), negq(d1), r0, r1);
    }
#endif

    // r1 = a^2, r2 = c^2.  Device, inlined flavour: the cross products a_i a_j (i < j) are taken once, against a doubled limb
    // (a < 2^255: the doubling loses no bit) -- 36 multiply-accumulates per square instead of 64 (tools/gen_mont_mul.py).
    static FK_HD void sqr2(const Fp &a, const Fp &c, Fp &r1, Fp &r2) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FK_NO_SQR2)
        if constexpr (INL) {
            Fp ad, ae, cd, ce;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                ae.v[i] = a.v[i] << 1; ce.v[i] = c.v[i] << 1;
                ad.v[i] = ae.v[i] | (i ? a.v[i - 1] >> 31 : 0); cd.v[i] = ce.v[i] | (i ? c.v[i - 1] >> 31 : 0);
            }
            sqr2_body_asm(a, ad, ae, c, cd, ce, r1, r2);
            return;
        }
#endif
        mul2(a, a, c, c, r1, r2);
    }

    // a0 b0 + a1 b1 + a2 b2 + a3 b3, canonical operands.  Device, inlined flavour: one Montgomery reduction for the four products.
    static FK_HD Fp dot4(const Fp &a0, const Fp &b0, const Fp &a1, const Fp &b1, const Fp &a2, const Fp &b2, const Fp &a3, const Fp &b3) {
        static_assert(!P::LAZY, "dot4: the bound holds for canonical operands");
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FK_NO_DOT4)
        if constexpr (INL) return dot4_body_asm(a0, b0, a1, b1, a2, b2, a3, b3);
#endif
        Fp x, y, u, v; mul2(a0, b0, a1, b1, x, y); mul2(a2, b2, a3, b3, u, v);
        return add(add(x, y), add(u, v));
    }

    // a b - c d.  Device, inlined flavour: ONE Montgomery reduction for the sum a b + (q - c) d (mulsum_body_asm), i.e. a
    // reduction, a subtraction and their carry handling less than two products and a difference.
    static FK_HD Fp mulsub(const Fp &a, const Fp &b, const Fp &c, const Fp &d) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FK_NO_MULSUM)
        if constexpr (INL) return mulsum_body_asm(a, b, negq(c), d);
#endif
        Fp x, y; mul2(a, b, c, d, x, y);
        return sub(x, y);
    }

    // CIOS Montgomery product a * b * 2^-256 mod p.
    static FK_HD Fp mul_body(const Fp &a, const Fp &b) {
        uint32_t t[8];
#pragma unroll
        for (int i = 0; i < 8; i++) t[i] = 0;
        uint32_t t8 = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t c = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                uint64_t x = (uint64_t)a.v[j] * b.v[i] + t[j] + c;
                t[j] = (uint32_t)x; c = x >> 32;
            }
            uint64_t top = (uint64_t)t8 + c;          // fits 33 bits
            uint32_t m = t[0] * P::INV;
            uint64_t x = (uint64_t)m * P::p(0) + t[0];
            c = x >> 32;
#pragma unroll
            for (int j = 1; j < 8; j++) {
                x = (uint64_t)m * P::p(j) + t[j] + c;
                t[j - 1] = (uint32_t)x; c = x >> 32;
            }
            top += c;
            t[7] = (uint32_t)top; t8 = (uint32_t)(top >> 32);
        }
        // a, b < 2p and 4p < 2^256 keep the running value < 2p, so t8 == 0 here
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = t[i];
        return fin(r);
    }
    static FK_HD Fp sqr(const Fp &a) { return mul(a, a); }

    // Montgomery -> canonical integer (multiply by 1)
    static FK_HD Fp from_mont(const Fp &a) {
        Fp o = zero(); o.v[0] = 1;
        return mul(a, o);
    }
    static FK_HD Fp to_mont(const Fp &a) { return mul(a, r2()); }

    // a^e, e given as 8 x u32 canonical LE (host-side helper; also fine on device)
    static FK_HD Fp pow(const Fp &a, const uint32_t e[8]) {
        Fp acc = one(), base = a;
        for (int i = 0; i < 256; i++) {
            if ((e[i >> 5] >> (i & 31)) & 1) acc = mul(acc, base);
            base = sqr(base);
        }
        return acc;
    }
    static FK_HD Fp pow_u64(const Fp &a, uint64_t e) {
        Fp acc = one(), base = a;
        while (e) { if (e & 1) acc = mul(acc, base); base = sqr(base); e >>= 1; }
        return acc;
    }
    static FK_HD Fp inv(const Fp &a) {  // Fermat; callers never pass zero where it matters
        uint32_t e[8]; uint32_t br = 2;
        for (int i = 0; i < 8; i++) {
            uint64_t t = (uint64_t)P::p(i) - br; e[i] = (uint32_t)t; br = (uint32_t)(t >> 63);
        }
        return pow(a, e);
    }
    static FK_HD Fp from_u64(uint64_t x) {
        Fp o = zero(); o.v[0] = (uint32_t)x; o.v[1] = (uint32_t)(x >> 32);
        return to_mont(o);
    }
};

#if defined(__HIP_DEVICE_COMPILE__)
template <class P>
__device__ __noinline__ Fp<P, false> mont_mul_call(Fp<P, false> a, Fp<P, false> b) {
    return Fp<P, false>::mul_body_asm(a, b);
}
#endif

struct FrLazyParams : FrParams {
    using Base = FrParams;
    static constexpr bool LAZY = true;
    static constexpr FK_HD uint32_t q(int i) { constexpr uint32_t t[8] = FK_FR_2P; return t[i]; }
};
using Fq = Fp<FqParams, true>;
using FqC = Fp<FqParams, false>;    // "cold": out-of-line multiply
using FqL = Fp<FqLazyParams, true>; // lazily reduced [0, 2p): bucket accumulators only
using FrL = Fp<FrLazyParams, true>; // the same for Fr: the butterflies of a transform pass
using Fr = Fp<FrParams, true>;

// Fq2 = Fq[u]/(u^2 + 1)   (pairing_ce bn256 tower)
template <class Fq>
struct alignas(16) Fq2T {
    using Fq2 = Fq2T;
    Fq c0, c1;
    static FK_HD Fq2 zero() { return Fq2{Fq::zero(), Fq::zero()}; }
    static FK_HD Fq2 one() { return Fq2{Fq::one(), Fq::zero()}; }
    FK_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    friend FK_HD bool operator==(const Fq2 &a, const Fq2 &b) { return a.c0 == b.c0 && a.c1 == b.c1; }
    friend FK_HD bool operator!=(const Fq2 &a, const Fq2 &b) { return !(a == b); }
    // the two components are independent: every addition / subtraction is one dual-chain operation
    static FK_HD Fq2 add(const Fq2 &a, const Fq2 &b) { Fq2 r; Fq::add2(a.c0, b.c0, a.c1, b.c1, r.c0, r.c1); return r; }
    static FK_HD Fq2 sub(const Fq2 &a, const Fq2 &b) { Fq2 r; Fq::sub2(a.c0, b.c0, a.c1, b.c1, r.c0, r.c1); return r; }
    static FK_HD Fq2 dbl(const Fq2 &a) { return add(a, a); }
    static FK_HD Fq2 neg(const Fq2 &a) { return sub(zero(), a); }
    // two independent Fq2 operations: the four component chains of each step run as two dual-chain operations
    static FK_HD void add2(const Fq2 &a, const Fq2 &b, const Fq2 &c, const Fq2 &d, Fq2 &r1, Fq2 &r2) { const Fq2 t = add(a, b); r2 = add(c, d); r1 = t; }
    static FK_HD void sub2(const Fq2 &a, const Fq2 &b, const Fq2 &c, const Fq2 &d, Fq2 &r1, Fq2 &r2) { const Fq2 t = sub(a, b); r2 = sub(c, d); r1 = t; }
    static FK_HD void addsub2(const Fq2 &a, const Fq2 &b, const Fq2 &c, const Fq2 &d, Fq2 &r1, Fq2 &r2) { const Fq2 t = add(a, b); r2 = sub(c, d); r1 = t; }
    static FK_HD Fq2 mul(const Fq2 &a, const Fq2 &b) {  // Karatsuba, 3 base multiplications -- or the fused schoolbook form
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (Fq::HAS_FQ2MUL) { Fq2 r; Fq::fq2mul(a.c0, a.c1, b.c0, b.c1, r.c0, r.c1); return r; }
#endif
        Fq aa, bb, sa, sb;
        Fq::mul2(a.c0, b.c0, a.c1, b.c1, aa, bb);
        Fq::add2(a.c0, a.c1, b.c0, b.c1, sa, sb);
        Fq t = Fq::mul(sa, sb);
        Fq u, r0;
        Fq::addsub2(aa, bb, aa, bb, u, r0);          // u = aa + bb, r0 = aa - bb
        return Fq2{r0, Fq::sub(t, u)};
    }
    // r1 = a*b, r2 = c*d.  Register pressure decides here (XYZZ<Fq2> is 64 registers of state): the two products
    // run one after the other; inside each, a0*b0 and a1*b1 form one dual chain (see mul).
    static FK_HD void mul2(const Fq2 &a, const Fq2 &b, const Fq2 &c, const Fq2 &d, Fq2 &r1, Fq2 &r2) {
        const Fq2 t = mul(a, b);
        r2 = mul(c, d);
        r1 = t;
    }
    static FK_HD Fq2 mulsub(const Fq2 &a, const Fq2 &b, const Fq2 &c, const Fq2 &d) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FK_NO_FQ2MULSUB)
        if constexpr (Fq::HAS_FQ2MUL) { Fq2 r; Fq::fq2mulsub(a.c0, a.c1, b.c0, b.c1, c.c0, c.c1, d.c0, d.c1, r.c0, r.c1); return r; }
#endif
        return sub(mul(a, b), mul(c, d));
    }
    static FK_HD void sqr2(const Fq2 &a, const Fq2 &c, Fq2 &r1, Fq2 &r2) { const Fq2 t = sqr(a); r2 = sqr(c); r1 = t; }
    static FK_HD Fq2 sqr(const Fq2 &a) {  // (c0+c1)(c0-c1), 2 c0 c1
        Fq s, d, m, n;
        Fq::addsub2(a.c0, a.c1, a.c0, a.c1, s, d);
        Fq::mul2(a.c0, a.c1, s, d, m, n);
        return Fq2{n, Fq::dbl(m)};
    }
    static FK_HD Fq2 inv(const Fq2 &a) {
        Fq n = Fq::inv(Fq::add(Fq::sqr(a.c0), Fq::sqr(a.c1)));
        return Fq2{Fq::mul(a.c0, n), Fq::neg(Fq::mul(a.c1, n))};
    }
};

using Fq2 = Fq2T<Fq>;
using Fq2C = Fq2T<FqC>;

// lazily reduced twin of a coordinate field (same layout; a canonical value is a valid lazy one)
template <class F> struct LazyOf;
template <> struct LazyOf<Fq> { using type = FqL; };
template <> struct LazyOf<Fq2T<Fq>> { using type = Fq2T<FqL>; };
template <class LP>
static FK_HD Fp<typename LP::Base, true> canon(const Fp<LP, true> &a) { Fp<typename LP::Base, true> r; for (int i = 0; i < 8; i++) r.v[i] = a.v[i]; return Fp<typename LP::Base, true>::reduce_once(r); }
template <class LP>
static FK_HD Fp<LP, true> lazy_of(const Fp<typename LP::Base, true> &a) { Fp<LP, true> r; for (int i = 0; i < 8; i++) r.v[i] = a.v[i]; return r; }
static FK_HD Fq2T<Fq> canon(const Fq2T<FqL> &a) { return Fq2T<Fq>{canon(a.c0), canon(a.c1)}; }

// cold twin of a coordinate field (same layout, out-of-line multiply)
template <class F> struct ColdOf;
template <> struct ColdOf<Fq> { using type = FqC; };
template <> struct ColdOf<Fq2> { using type = Fq2C; };

}  // namespace fk
