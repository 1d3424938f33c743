// BN254 G1 / G2 group arithmetic, generic over the coordinate field (Fq or Fq2).
//
// Affine points use the raw layout fawkes-crypto exchanges with bellman (`into_raw_uncompressed_le`,
// backend/bellman_groth16/group.rs:57-66 and :97-103): Montgomery LE x || y, and the all-zero buffer is
// the point at infinity (group.rs:55,71-72,89-93).  Accumulators are XYZZ (x = X/ZZ, y = Y/ZZZ,
// ZZ^3 = ZZZ^2; infinity <=> ZZ == 0): a mixed addition costs 8M+2S instead of Jacobian's 7M+4S and
// needs no inversion.  Any correct group law yields the same affine result, so the 256 proof bytes
// are independent of the coordinate system.
#pragma once
#include "field.hpp"

namespace fk {

template <class F>
struct alignas(16) Affine {
    F x, y;
    FK_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
    static FK_HD Affine inf() { return Affine{F::zero(), F::zero()}; }
};

template <class F>
struct alignas(16) Xyzz {
    F x, y, zz, zzz;

    static FK_HD Xyzz inf() { return Xyzz{F::zero(), F::zero(), F::zero(), F::zero()}; }
    FK_HD bool is_inf() const { return zz.is_zero(); }
    static FK_HD Xyzz from_affine(const Affine<F> &p) {
        if (p.is_inf()) return inf();
        return Xyzz{p.x, p.y, F::one(), F::one()};
    }

    // 2 * (affine p), p not infinity (mdbl-2008-s-1)
    static FK_HD Xyzz dbl_affine(const Affine<F> &p) {
        F u = F::dbl(p.y);
        F v = F::sqr(u);
        F w = F::mul(u, v);
        F s = F::mul(p.x, v);
        F xx = F::sqr(p.x);
        F m = F::add(F::dbl(xx), xx);
        Xyzz r;
        r.x = F::sub(F::sqr(m), F::dbl(s));
        r.y = F::sub(F::mul(m, F::sub(s, r.x)), F::mul(w, p.y));
        r.zz = v;
        r.zzz = w;
        return r;
    }

    // dbl-2008-s-1
    static FK_HD Xyzz dbl(const Xyzz &p) {
        if (p.is_inf()) return p;
        F u = F::dbl(p.y);
        F v = F::sqr(u);
        F w = F::mul(u, v);
        F s = F::mul(p.x, v);
        F xx = F::sqr(p.x);
        F m = F::add(F::dbl(xx), xx);
        Xyzz r;
        r.x = F::sub(F::sqr(m), F::dbl(s));
        r.y = F::sub(F::mul(m, F::sub(s, r.x)), F::mul(w, p.y));
        r.zz = F::mul(v, p.zz);
        r.zzz = F::mul(w, p.zzz);
        return r;
    }

    // acc += q (affine), madd-2008-s with the exceptional cases handled
    FK_HD void add_mixed(const Affine<F> &q) {
        if (q.is_inf()) return;
        add_mixed_nz(q);
    }
    // the same for a q known not to be the point at infinity
    FK_HD void add_mixed_nz(const Affine<F> &q) {
        if (is_inf()) { x = q.x; y = q.y; zz = F::one(); zzz = F::one(); return; }
        // the ten multiplications form five independent pairs -> four dual-chain products (F::mul2) and one fused difference of two (F::mulsub)
        F u2, s2;
        F::mul2(q.x, zz, q.y, zzz, u2, s2);
        F p, r;
        F::sub2(u2, x, s2, y, p, r);                 // independent differences: one dual-chain operation
        if (p.is_zero()) {
            if (r.is_zero()) *this = dbl_affine(q); else *this = inf();
            return;
        }
        F pp, rr;
        F::sqr2(p, r, pp, rr);
        F ppp, q_;
        F::mul2(p, pp, x, pp, ppp, q_);
        F q2, t1;
        F::addsub2(q_, q_, rr, ppp, q2, t1);         // q2 = 2 q, t1 = rr - ppp
        F x3 = F::sub(t1, q2);
        y = F::mulsub(r, F::sub(q_, x3), y, ppp);      // r (q - x3) - y ppp: one reduction in G1 on the device (field.hpp)
        x = x3;
        F::mul2(zz, pp, zzz, ppp, zz, zzz);
    }

    // acc += q (XYZZ), add-2008-s with the exceptional cases handled
    FK_HD void add(const Xyzz &q) {
        if (q.is_inf()) return;
        if (is_inf()) { *this = q; return; }
        F u1 = F::mul(x, q.zz);
        F u2 = F::mul(q.x, zz);
        F s1 = F::mul(y, q.zzz);
        F s2 = F::mul(q.y, zzz);
        F p = F::sub(u2, u1);
        F r = F::sub(s2, s1);
        if (p.is_zero()) {
            if (r.is_zero()) *this = dbl(*this); else *this = inf();
            return;
        }
        F pp = F::sqr(p);
        F ppp = F::mul(p, pp);
        F q_ = F::mul(u1, pp);
        F x3 = F::sub(F::sub(F::sqr(r), ppp), F::dbl(q_));
        y = F::sub(F::mul(r, F::sub(q_, x3)), F::mul(s1, ppp));
        x = x3;
        zz = F::mul(F::mul(zz, q.zz), pp);
        zzz = F::mul(F::mul(zzz, q.zzz), ppp);
    }

    FK_HD Affine<F> to_affine() const {  // one inversion: 1/(zz*zzz) -> 1/zz, 1/zzz
        if (is_inf()) return Affine<F>::inf();
        F t = F::inv(F::mul(zz, zzz));
        F izz = F::mul(t, zzz);
        F izzz = F::mul(t, zz);
        return Affine<F>{F::mul(x, izz), F::mul(y, izzz)};
    }

    // k * p, k canonical 8 x u32 LE (host-side assembly only)
    static FK_HD Xyzz mul_scalar(const Xyzz &p, const uint32_t k[8]) {
        Xyzz acc = inf();
        for (int i = 255; i >= 0; i--) {
            acc = dbl(acc);
            if ((k[i >> 5] >> (i & 31)) & 1) acc.add(p);
        }
        return acc;
    }
};

template <class F>
static FK_HD Affine<F> affine_neg_if(const Affine<F> &p, bool neg) {
    // -(x, y) = (x, -y); infinity (0,0) maps to itself because neg(0) == 0
    return neg ? Affine<F>{p.x, F::neg(p.y)} : p;
}

using G1Affine = Affine<Fq>;
using G2Affine = Affine<Fq2>;
using G1Xyzz = Xyzz<Fq>;
using G2Xyzz = Xyzz<Fq2>;

}  // namespace fk
