// Microbenchmark behind DESIGN.md section 3.3 ("batched-affine bucket accumulation: bounded experiment, killed"):
// G1 point additions per second IN REGISTERS -- the inner loop of the bucket accumulation without its memory traffic -- for
//   (a) the production form: XYZZ mixed addition, lazily reduced coordinates (curve.hpp add_mixed_nz over FqL), one accumulator
//       per lane, 4 waves per SIMD;
//   (b) batched AFFINE addition: K accumulators per lane, K independent additions per step sharing ONE inversion through
//       Montgomery's trick (3 products per addition for the trick, 2 products + 1 squaring for lambda, x3, y3), with
//         inv = 0: the inversion replaced by ONE product -- the bound of a FREE inversion (results are not points; the
//                  instruction mix and the register pressure are those of the real thing),
//         inv = 1: a real inversion per step (Fermat, Fq::inv) -- checked against (a) point by point.
// What (b) cannot avoid: K accumulators (16 registers each) + K differences + K prefix products live in registers, so K <= 4 at
// 2 waves per SIMD; amortising a >= 30-product inversion over >= 64 additions needs the accumulators in memory.
// Build: hipcc -O3 --offload-arch=gfx950 -I../../fawkes-crypto_amd/csrc addbench.hip -o addbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "curve.hpp"
using namespace fk;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MINW>
__global__ __launch_bounds__(256, MINW) void bench_xyzz(const G1Affine *pts, G1Xyzz *out, int iters) {
    using FL = FqL;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Affine<FL> q[4];
    for (int j = 0; j < 4; j++) { G1Affine p = pts[(i * 4 + j) & 1023]; __builtin_memcpy(&q[j], &p, sizeof p); }
    Xyzz<FL> acc = Xyzz<FL>::inf();
    for (int k = 0; k < iters; k++) { acc.add_mixed(q[0]); acc.add_mixed_nz(q[1]); acc.add_mixed_nz(q[2]); acc.add_mixed_nz(q[3]); }
    out[i] = acc.is_inf() ? G1Xyzz::inf() : G1Xyzz{canon(acc.x), canon(acc.y), canon(acc.zz), canon(acc.zzz)};
}

// K affine accumulators per lane; every step adds the point q[(k + j) & 3] to accumulator j
template <int K, int MINW, int INV>
__global__ __launch_bounds__(256, MINW) void bench_affine(const G1Affine *pts, G1Affine *out, int iters) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    G1Affine q[4];
    for (int j = 0; j < 4; j++) q[j] = pts[(i * 4 + j) & 1023];
    G1Affine acc[K];
#pragma unroll
    for (int j = 0; j < K; j++) acc[j] = pts[(i * 7 + 500 + j * 13) & 1023];          // distinct starting points (no doubling case below)
    for (int k = 0; k < iters; k++) {
        Fq d[K], pre[K];
#pragma unroll
        for (int j = 0; j < K; j++) {
            d[j] = Fq::sub(q[(k + j) & 3].x, acc[j].x);
            pre[j] = j ? Fq::mul(pre[j - 1], d[j]) : d[j];
        }
        Fq inv = INV ? Fq::inv(pre[K - 1]) : Fq::mul(pre[K - 1], pre[0]);
#pragma unroll
        for (int j = K - 1; j >= 0; j--) {
            const Fq dinv = j ? Fq::mul(inv, pre[j - 1]) : inv;
            if (j) inv = Fq::mul(inv, d[j]);
            const G1Affine &p = q[(k + j) & 3];
            const Fq lam = Fq::mul(Fq::sub(p.y, acc[j].y), dinv);
            const Fq x3 = Fq::sub(Fq::sub(Fq::sqr(lam), acc[j].x), p.x);
            acc[j].y = Fq::sub(Fq::mul(lam, Fq::sub(acc[j].x, x3)), acc[j].y);
            acc[j].x = x3;
        }
    }
#pragma unroll
    for (int j = 0; j < K; j++) out[i * K + j] = acc[j];
}

int main() {
    const int blocks = 256 * 8, threads = 256, iters = 64;
    const size_t n = (size_t)blocks * threads;
    std::vector<G1Affine> h(1024);
    G1Affine g; g.x = Fq::from_u64(1); g.y = Fq::from_u64(2);
    G1Xyzz cur = G1Xyzz::from_affine(g);
    for (int i = 0; i < 1024; i++) { h[i] = cur.to_affine(); cur.add_mixed(g); if (i % 7 == 3) cur = G1Xyzz::dbl(cur); }
    G1Affine *dp, *da; G1Xyzz *dx;
    CK(hipMalloc(&dp, 1024 * sizeof(G1Affine))); CK(hipMalloc(&dx, n * sizeof(G1Xyzz))); CK(hipMalloc(&da, n * 8 * sizeof(G1Affine)));
    CK(hipMemcpy(dp, h.data(), 1024 * sizeof(G1Affine), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto launch) { launch(); hipDeviceSynchronize(); hipEventRecord(e0); for (int r = 0; r < 3; r++) launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 3 * 1e-3; };
    const double tx = timeit([&] { hipLaunchKernelGGL(bench_xyzz<4>, dim3(blocks), dim3(threads), 0, 0, dp, dx, iters); });
    const double base = (double)n * iters * 4 / tx / 1e9;
    printf("XYZZ mixed addition, lazily reduced, 1 accumulator / lane, 4 waves/SIMD : %6.2f G additions/s  (1.00)\n", base);
#define AFF(K_, W_, I_, label) { const double t = timeit([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(bench_affine<K_, W_, I_>), dim3(blocks), dim3(threads), 0, 0, dp, da, iters); }); \
        const double r = (double)n * iters * K_ / t / 1e9; printf("batched affine, K = %d / lane, %d waves/SIMD, %-28s : %6.2f G additions/s  (%.2f)\n", K_, W_, label, r, r / base); }
    AFF(2, 4, 0, "FREE inversion (bound)");
    AFF(4, 2, 0, "FREE inversion (bound)");
    AFF(4, 3, 0, "FREE inversion (bound)");
    AFF(8, 1, 0, "FREE inversion (bound)");
    AFF(4, 2, 1, "Fermat inversion per step");
    AFF(8, 1, 1, "Fermat inversion per step");
    // correctness of the affine formulas (K = 4, real inversion): accumulator j of lane i after the loop, against host XYZZ arithmetic
    hipLaunchKernelGGL(HIP_KERNEL_NAME(bench_affine<4, 2, 1>), dim3(blocks), dim3(threads), 0, 0, dp, da, 5);
    std::vector<G1Affine> got(n * 4);
    CK(hipMemcpy(got.data(), da, n * 4 * sizeof(G1Affine), hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < n; i += 4099)
        for (int j = 0; j < 4; j++) {
            G1Xyzz a = G1Xyzz::from_affine(h[(i * 7 + 500 + j * 13) & 1023]);
            for (int k = 0; k < 5; k++) a.add_mixed(h[(i * 4 + ((k + j) & 3)) & 1023]);
            const G1Affine w = a.to_affine();
            bad += !(w.x == got[i * 4 + j].x && w.y == got[i * 4 + j].y);
        }
    printf("batched-affine results equal the XYZZ sums: %s\n", bad ? "NO" : "yes");
    return 0;
}
