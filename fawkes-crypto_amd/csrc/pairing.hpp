// BN254 pairing for the verifier half of the backend:
//   /root/reference/fawkes-crypto/src/backend/bellman_groth16/verifier.rs:75-81  verify(vk, proof, inputs)
//   -> bellman_ce::groth16::{prepare_verifying_key, verify_proof}, un-vendored.
//
// Generic over the base field type, so the same code is the host verifier (fk_verify: Fq, CIOS on the CPU) and one lane of
// the batch verifier kernel (fk_verify_batch_dev: FqC, out-of-line multiply).  Tower Fq2 = Fq[u]/(u^2+1),
// Fq6 = Fq2[v]/(v^3 - xi), xi = 9 + u, Fq12 = Fq6[w]/(w^2 - v); G2 is the D-type sextic twist y^2 = x^3 + 3/xi.
//
// The pairing computed is the ATE pairing f_{T,Q}(P)^((p^12-1)/r) with T = t - 1 = 6 x^2 (127 bits), not bellman's optimal
// ate: a verifier only needs SOME non-degenerate bilinear pairing used consistently on both sides of
//     e(A, B) = e(alpha, beta) e(sum x_i ic_i, gamma) e(C, delta),
// and this one needs no Frobenius constants -- affine line functions (one Fq2 inversion per step), plain square-and-multiply
// for the final exponentiation.  Accept / reject is identical to bellman's; checked against the independent big-int
// verifier of the oracle (tests/test_verify.py).  It is not a hot path: ~25 ms per proof on one host core.
#pragma once
#include "curve.hpp"

namespace fk {

template <class Fq>
struct Fq6T {
    using F2 = Fq2T<Fq>;
    F2 c0, c1, c2;
    static FK_HD Fq6T zero() { return Fq6T{F2::zero(), F2::zero(), F2::zero()}; }
    static FK_HD Fq6T one() { return Fq6T{F2::one(), F2::zero(), F2::zero()}; }
    FK_HD bool is_zero() const { return c0.is_zero() && c1.is_zero() && c2.is_zero(); }
    static FK_HD F2 mul_xi(const F2 &a) {          // (9 + u) a
        Fq t0 = Fq::dbl(Fq::dbl(Fq::dbl(a.c0))), t1 = Fq::dbl(Fq::dbl(Fq::dbl(a.c1)));
        t0 = Fq::add(t0, a.c0); t1 = Fq::add(t1, a.c1);               // 9 a0, 9 a1
        return F2{Fq::sub(t0, a.c1), Fq::add(t1, a.c0)};
    }
    static FK_HD Fq6T add(const Fq6T &a, const Fq6T &b) { return Fq6T{F2::add(a.c0, b.c0), F2::add(a.c1, b.c1), F2::add(a.c2, b.c2)}; }
    static FK_HD Fq6T sub(const Fq6T &a, const Fq6T &b) { return Fq6T{F2::sub(a.c0, b.c0), F2::sub(a.c1, b.c1), F2::sub(a.c2, b.c2)}; }
    static FK_HD Fq6T neg(const Fq6T &a) { return Fq6T{F2::neg(a.c0), F2::neg(a.c1), F2::neg(a.c2)}; }
    static FK_HD Fq6T mul(const Fq6T &a, const Fq6T &b) {             // v^3 = xi
        const F2 a0b0 = F2::mul(a.c0, b.c0), a1b1 = F2::mul(a.c1, b.c1), a2b2 = F2::mul(a.c2, b.c2);
        const F2 t12 = F2::sub(F2::sub(F2::mul(F2::add(a.c1, a.c2), F2::add(b.c1, b.c2)), a1b1), a2b2);   // a1b2 + a2b1
        const F2 t01 = F2::sub(F2::sub(F2::mul(F2::add(a.c0, a.c1), F2::add(b.c0, b.c1)), a0b0), a1b1);   // a0b1 + a1b0
        const F2 t02 = F2::sub(F2::sub(F2::mul(F2::add(a.c0, a.c2), F2::add(b.c0, b.c2)), a0b0), a2b2);   // a0b2 + a2b0
        return Fq6T{F2::add(a0b0, mul_xi(t12)), F2::add(t01, mul_xi(a2b2)), F2::add(t02, a1b1)};
    }
    static FK_HD Fq6T mul_v(const Fq6T &a) { return Fq6T{mul_xi(a.c2), a.c0, a.c1}; }
    static FK_HD Fq6T inv(const Fq6T &a) {
        const F2 t0 = F2::sub(F2::sqr(a.c0), mul_xi(F2::mul(a.c1, a.c2)));
        const F2 t1 = F2::sub(mul_xi(F2::sqr(a.c2)), F2::mul(a.c0, a.c1));
        const F2 t2 = F2::sub(F2::sqr(a.c1), F2::mul(a.c0, a.c2));
        const F2 d = F2::add(F2::mul(a.c0, t0), mul_xi(F2::add(F2::mul(a.c2, t1), F2::mul(a.c1, t2))));
        const F2 di = F2::inv(d);
        return Fq6T{F2::mul(t0, di), F2::mul(t1, di), F2::mul(t2, di)};
    }
};

template <class Fq>
struct Fq12T {
    using F6 = Fq6T<Fq>;
    using F2 = Fq2T<Fq>;
    F6 c0, c1;
    static FK_HD Fq12T one() { return Fq12T{F6::one(), F6::zero()}; }
    FK_HD bool is_one() const { return c1.is_zero() && c0.c1.is_zero() && c0.c2.is_zero() && c0.c0.c1.is_zero() && c0.c0.c0 == Fq::one(); }
    static FK_HD Fq12T mul(const Fq12T &a, const Fq12T &b) {          // w^2 = v
        const F6 aa = F6::mul(a.c0, b.c0), bb = F6::mul(a.c1, b.c1);
        const F6 cross = F6::sub(F6::sub(F6::mul(F6::add(a.c0, a.c1), F6::add(b.c0, b.c1)), aa), bb);
        return Fq12T{F6::add(aa, F6::mul_v(bb)), cross};
    }
    static FK_HD Fq12T sqr(const Fq12T &a) { return mul(a, a); }
    static FK_HD Fq12T conj(const Fq12T &a) { return Fq12T{a.c0, F6::neg(a.c1)}; }     // a^(p^6)
    static FK_HD Fq12T inv(const Fq12T &a) {
        const F6 d = F6::inv(F6::sub(F6::mul(a.c0, a.c0), F6::mul_v(F6::mul(a.c1, a.c1))));
        return Fq12T{F6::mul(a.c0, d), F6::neg(F6::mul(a.c1, d))};
    }
    // a^e, e little-endian 32-bit words
    static FK_HD Fq12T pow(const Fq12T &a, const uint32_t *e, int nwords) {
        Fq12T acc = one();
        bool started = false;
        for (int i = nwords * 32 - 1; i >= 0; i--) {
            if (started) acc = sqr(acc);
            if ((e[i >> 5] >> (i & 31)) & 1) { acc = started ? mul(acc, a) : a; started = true; }
        }
        return acc;
    }
};

// one Miller loop f_{T,Q}(P) of the ate pairing, T = t - 1 = 6 x^2.  P in G1 (affine), Q on the twist (affine).
template <class Fq>
static FK_HD Fq12T<Fq> miller_loop(const Affine<Fq> &P, const Affine<Fq2T<Fq>> &Q) {
    using F2 = Fq2T<Fq>; using F6 = Fq6T<Fq>; using F12 = Fq12T<Fq>;
    F12 f = F12::one();
    if (P.is_inf() || Q.is_inf()) return f;
    const uint32_t T[4] = FK_ATE_LOOP_T;            // 127 bits
    F2 xr = Q.x, yr = Q.y;
    // the line through psi(R) with slope lambda w, at P:  yP - lambda xP w + (lambda xR - yR) w^3   (w^2 = v, w^3 = v w)
    auto line = [&](const F2 &lam, const F2 &x0, const F2 &y0) {
        F12 l;
        l.c0 = F6{F2{P.y, Fq::zero()}, F2::zero(), F2::zero()};
        const F2 lx = F2{Fq::mul(lam.c0, P.x), Fq::mul(lam.c1, P.x)};
        l.c1 = F6{F2::neg(lx), F2::sub(F2::mul(lam, x0), y0), F2::zero()};
        return l;
    };
    int top = 127;
    while (!((T[top >> 5] >> (top & 31)) & 1)) top--;
    for (int i = top - 1; i >= 0; i--) {
        // doubling step: lambda = 3 x^2 / (2 y)
        const F2 xx = F2::sqr(xr);
        const F2 lam = F2::mul(F2::add(F2::dbl(xx), xx), F2::inv(F2::dbl(yr)));
        f = F12::mul(F12::sqr(f), line(lam, xr, yr));
        const F2 x3 = F2::sub(F2::sqr(lam), F2::dbl(xr));
        yr = F2::sub(F2::mul(lam, F2::sub(xr, x3)), yr);
        xr = x3;
        if ((T[i >> 5] >> (i & 31)) & 1) {          // addition step with Q (R != +-Q for a point of prime order r > T)
            const F2 lam2 = F2::mul(F2::sub(yr, Q.y), F2::inv(F2::sub(xr, Q.x)));
            f = F12::mul(f, line(lam2, xr, yr));
            const F2 x4 = F2::sub(F2::sub(F2::sqr(lam2), xr), Q.x);
            yr = F2::sub(F2::mul(lam2, F2::sub(xr, x4)), yr);
            xr = x4;
        }
    }
    return f;
}

// f^((p^12 - 1) / r) = ((conj(f) / f)^(p^2 + 1))^((p^4 - p^2 + 1) / r)
template <class Fq>
static FK_HD Fq12T<Fq> final_exponentiation(const Fq12T<Fq> &f) {
    using F12 = Fq12T<Fq>;
    const uint32_t e1[16] = FK_FEXP_P2_PLUS_1, e2[24] = FK_FEXP_HARD;
    const F12 g = F12::mul(F12::conj(f), F12::inv(f));
    return F12::pow(F12::pow(g, e1, 16), e2, 24);
}

// e(A, B) e(-alpha, beta) e(-acc, gamma) e(-C, delta) == 1
template <class Fq>
static FK_HD bool groth16_check(const Affine<Fq> &A, const Affine<Fq2T<Fq>> &B, const Affine<Fq> &C, const Affine<Fq> &alpha,
                                const Affine<Fq2T<Fq>> &beta, const Affine<Fq2T<Fq>> &gamma, const Affine<Fq2T<Fq>> &delta, const Affine<Fq> &acc) {
    using F12 = Fq12T<Fq>;
    auto negp = [](const Affine<Fq> &p) { return p.is_inf() ? p : Affine<Fq>{p.x, Fq::neg(p.y)}; };
    F12 m = miller_loop<Fq>(A, B);
    m = F12::mul(m, miller_loop<Fq>(negp(alpha), beta));
    m = F12::mul(m, miller_loop<Fq>(negp(acc), gamma));
    m = F12::mul(m, miller_loop<Fq>(negp(C), delta));
    return final_exponentiation<Fq>(m).is_one();
}

}  // namespace fk
