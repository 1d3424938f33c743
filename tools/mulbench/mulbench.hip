// Microbenchmark: BN254 Montgomery multiplication variants on gfx950 (throughput in G mul/s) and raw
// instruction-rate calibration.  Build: hipcc -O3 --offload-arch=gfx950 -I../../fawkes-crypto_amd/csrc mulbench.hip -o mulbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "field.hpp"
using namespace fk;

// ---- V1: product scanning, 96-bit accumulator, carry-out of v_mad_u64_u32 consumed by v_addc
template <class P>
struct V1 {
    static __device__ __forceinline__ void mac(uint64_t &lo, uint32_t &hi, uint32_t a, uint32_t b) {
        asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
    }
    static __device__ __forceinline__ void mac_c(uint64_t &lo, uint32_t &hi, uint32_t a, uint32_t c) {  // c: constant -> SGPR
        asm("v_mad_u64_u32 %0, vcc, %3, %2, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "s"(c) : "vcc");
    }
    static __device__ __forceinline__ Fp<P> mul(const Fp<P> &a, const Fp<P> &b) {
        uint64_t lo = 0; uint32_t hi = 0;
        uint32_t m[8]; Fp<P> r;
#pragma unroll
        for (int k = 0; k < 8; k++) {
#pragma unroll
            for (int i = 0; i < k; i++) mac_c(lo, hi, m[i], P::p(k - i));
#pragma unroll
            for (int i = 0; i <= k; i++) mac(lo, hi, a.v[i], b.v[k - i]);
            m[k] = (uint32_t)lo * P::INV;
            mac_c(lo, hi, m[k], P::p(0));
            lo = (lo >> 32) | ((uint64_t)hi << 32); hi = 0;
        }
#pragma unroll
        for (int k = 8; k < 16; k++) {
#pragma unroll
            for (int i = k - 7; i < 8; i++) { mac(lo, hi, a.v[i], b.v[k - i]); mac_c(lo, hi, m[i], P::p(k - i)); }
            if (k < 15 || true) { r.v[k - 8] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32); hi = 0; }
        }
        return Fp<P>::reduce_once(r);
    }
};

// ---- V2: as V1 but separate asm statements (scheduler may interleave independent chains): two accumulators
template <class P>
struct V2 {
    static __device__ __forceinline__ void mac(uint64_t &lo, uint32_t &hi, uint32_t a, uint32_t b) {
        uint64_t c;
        asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(lo), "=s"(c) : "v"(a), "v"(b));
        asm("v_addc_co_u32 %0, %1, 0, %0, %1" : "+v"(hi), "+s"(c));
    }
    static __device__ __forceinline__ void mac_c(uint64_t &lo, uint32_t &hi, uint32_t a, uint32_t cst) {
        uint64_t c;
        asm("v_mad_u64_u32 %0, %1, %3, %2, %0" : "+v"(lo), "=s"(c) : "v"(a), "s"(cst));
        asm("v_addc_co_u32 %0, %1, 0, %0, %1" : "+v"(hi), "+s"(c));
    }
    static __device__ __forceinline__ Fp<P> mul(const Fp<P> &a, const Fp<P> &b) {
        uint64_t lo = 0, lo2 = 0; uint32_t hi = 0, hi2 = 0;
        uint32_t m[8]; Fp<P> r;
#pragma unroll
        for (int k = 0; k < 8; k++) {
#pragma unroll
            for (int i = 0; i < k; i++) mac_c(lo2, hi2, m[i], P::p(k - i));
#pragma unroll
            for (int i = 0; i <= k; i++) mac(lo, hi, a.v[i], b.v[k - i]);
            // merge chains
            { uint64_t s = lo + lo2; uint32_t cy = s < lo; lo = s; hi = hi + hi2 + cy; lo2 = 0; hi2 = 0; }
            m[k] = (uint32_t)lo * P::INV;
            mac_c(lo, hi, m[k], P::p(0));
            lo = (lo >> 32) | ((uint64_t)hi << 32); hi = 0;
        }
#pragma unroll
        for (int k = 8; k < 16; k++) {
#pragma unroll
            for (int i = k - 7; i < 8; i++) { mac(lo, hi, a.v[i], b.v[k - i]); mac_c(lo2, hi2, m[i], P::p(k - i)); }
            { uint64_t s = lo + lo2; uint32_t cy = s < lo; lo = s; hi = hi + hi2 + cy; lo2 = 0; hi2 = 0; }
            r.v[k - 8] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32); hi = 0;
        }
        return Fp<P>::reduce_once(r);
    }
};

struct V0 { static __device__ __forceinline__ Fq mul(const Fq &a, const Fq &b) { return Fq::mul_body(a, b); } };   // CIOS C loop
struct V3 { static __device__ __forceinline__ Fq mul(const Fq &a, const Fq &b) { return Fq::mul(a, b); } };        // production (generated asm)
struct V4 { static __device__ __forceinline__ Fq mul(const Fq &a, const Fq &b) {                                  // production, out-of-line call
    FqC x, y; for (int i = 0; i < 8; i++) { x.v[i] = a.v[i]; y.v[i] = b.v[i]; }
    FqC z = FqC::mul(x, y); Fq r; for (int i = 0; i < 8; i++) r.v[i] = z.v[i]; return r; } };

// two independent chains per lane: (a,b) and (c,d); single-product version does them one after the other,
// the dual version uses the interleaved mul2
template <bool DUAL>
__global__ __launch_bounds__(256) void bench2_kernel(const Fq *in, Fq *out, int iters) {
    extern __shared__ uint32_t occupancy_pad[];
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fq a = in[2 * i], b = in[2 * i + 1], c = Fq::add(a, b), d = Fq::sub(a, b);
    for (int k = 0; k < iters; k++) {
        if (DUAL) { Fq x, y; Fq::mul2(a, b, c, d, x, y); a = x; c = y; Fq::mul2(b, a, d, c, x, y); b = x; d = y; }
        else { a = Fq::mul(a, b); c = Fq::mul(c, d); b = Fq::mul(b, a); d = Fq::mul(d, c); }
    }
    if (occupancy_pad[0] == 12345) a = b;   // keep the LDS allocation alive
    out[i] = Fq::add(Fq::add(a, b), Fq::add(c, d));
}

template <class M>
__global__ __launch_bounds__(256) void bench_kernel(const Fq *in, Fq *out, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fq a = in[2 * i], b = in[2 * i + 1];
    for (int k = 0; k < iters; k++) { a = M::mul(a, b); b = M::mul(b, a); }
    out[i] = Fq::add(a, b);
}

// raw instruction rates
__global__ __launch_bounds__(256) void rate_mad(uint32_t *out, int iters) {
    uint32_t x = threadIdx.x + 1, y = blockIdx.x + 3; uint64_t a0 = x, a1 = y, a2 = x ^ y, a3 = x + y;
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y) : "vcc");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 + a1 + a2 + a3);
}
__global__ __launch_bounds__(256) void rate_add(uint32_t *out, int iters) {
    uint32_t x = threadIdx.x + 1, a0 = x, a1 = x * 3, a2 = x * 5, a3 = x * 7;
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
__global__ __launch_bounds__(256) void rate_addc(uint32_t *out, int iters) {
    uint32_t x = threadIdx.x + 1, a0 = x, a1 = x * 3, a2 = x * 5, a3 = x * 7;
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            asm volatile("v_add_co_u32 %0, vcc, %0, %4\n\tv_addc_co_u32 %1, vcc, %1, %4, vcc\n\tv_addc_co_u32 %2, vcc, %2, %4, vcc\n\tv_addc_co_u32 %3, vcc, %3, %4, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x) : "vcc");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
__global__ __launch_bounds__(256) void rate_mullo(uint32_t *out, int iters) {
    uint32_t x = threadIdx.x + 1, a0 = x, a1 = x * 3, a2 = x * 5, a3 = x * 7;
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            asm volatile("v_mul_lo_u32 %0, %0, %4\n\tv_mul_lo_u32 %1, %1, %4\n\tv_mul_lo_u32 %2, %2, %4\n\tv_mul_lo_u32 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
__global__ __launch_bounds__(256) void rate_dfma(double *out, int iters) {
    double x = threadIdx.x + 1.5, a0 = x, a1 = x * 3, a2 = x * 5, a3 = x * 7;
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            asm volatile("v_fma_f64 %0, %0, %4, %4\n\tv_fma_f64 %1, %1, %4, %4\n\tv_fma_f64 %2, %2, %4, %4\n\tv_fma_f64 %3, %3, %4, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
__global__ __launch_bounds__(256) void rate_mad24(uint32_t *out, int iters) {
    uint32_t x = threadIdx.x + 1, a0 = x, a1 = x * 3, a2 = x * 5, a3 = x * 7;
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            asm volatile("v_mad_u32_u24 %0, %0, %4, %4\n\tv_mad_u32_u24 %1, %1, %4, %4\n\tv_mad_u32_u24 %2, %2, %4, %4\n\tv_mad_u32_u24 %3, %3, %4, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <class K, class... A>
static double time_kernel(K k, dim3 g, dim3 b, int reps, A... args) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, g, b, 0, 0, args...); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k, g, b, 0, 0, args...);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps * 1e-3;
}

int main() {
    const int blocks = 256 * 8, threads = 256, iters = 200;
    const size_t n = (size_t)blocks * threads;
    std::vector<Fq> h(2 * n);
    uint64_t s = 12345;
    for (size_t i = 0; i < 2 * n; i++) {
        for (int k = 0; k < 8; k++) { s = s * 6364136223846793005ull + 1442695040888963407ull; h[i].v[k] = (uint32_t)(s >> 32); }
        h[i].v[7] &= 0x0fffffff;
    }
    Fq *din, *d0, *d1, *d2;
    CK(hipMalloc(&din, 2 * n * sizeof(Fq))); CK(hipMalloc(&d0, n * sizeof(Fq))); CK(hipMalloc(&d1, n * sizeof(Fq))); CK(hipMalloc(&d2, n * sizeof(Fq)));
    CK(hipMemcpy(din, h.data(), 2 * n * sizeof(Fq), hipMemcpyHostToDevice));
    double t0 = time_kernel(bench_kernel<V0>, dim3(blocks), dim3(threads), 3, (const Fq *)din, d0, iters);
    double t1 = time_kernel(bench_kernel<V1<FqParams>>, dim3(blocks), dim3(threads), 3, (const Fq *)din, d1, iters);
    double t2 = time_kernel(bench_kernel<V2<FqParams>>, dim3(blocks), dim3(threads), 3, (const Fq *)din, d2, iters);
    Fq *d3, *d4; CK(hipMalloc(&d3, n * sizeof(Fq))); CK(hipMalloc(&d4, n * sizeof(Fq)));
    double t3 = time_kernel(bench_kernel<V3>, dim3(blocks), dim3(threads), 3, (const Fq *)din, d3, iters);
    double t4 = time_kernel(bench_kernel<V4>, dim3(blocks), dim3(threads), 3, (const Fq *)din, d4, iters);
    std::vector<Fq> r3(n), r4(n);
    CK(hipMemcpy(r3.data(), d3, n * sizeof(Fq), hipMemcpyDeviceToHost));
    CK(hipMemcpy(r4.data(), d4, n * sizeof(Fq), hipMemcpyDeviceToHost));
    std::vector<Fq> r0(n), r1(n), r2(n);
    CK(hipMemcpy(r0.data(), d0, n * sizeof(Fq), hipMemcpyDeviceToHost));
    CK(hipMemcpy(r1.data(), d1, n * sizeof(Fq), hipMemcpyDeviceToHost));
    CK(hipMemcpy(r2.data(), d2, n * sizeof(Fq), hipMemcpyDeviceToHost));
    size_t bad1 = 0, bad2 = 0, bad3 = 0, bad4 = 0;
    for (size_t i = 0; i < n; i++) { bad1 += !(r0[i] == r1[i]); bad2 += !(r0[i] == r2[i]); bad3 += !(r0[i] == r3[i]); bad4 += !(r0[i] == r4[i]); }
    const double muls = (double)n * iters * 2;
    printf("V0 (CIOS C code)              : %.1f G mul/s\n", muls / t0 / 1e9);
    printf("V1 (product scan, asm carry)  : %.1f G mul/s   mismatches %zu\n", muls / t1 / 1e9, bad1);
    printf("V2 (two chains, split asm)    : %.1f G mul/s   mismatches %zu\n", muls / t2 / 1e9, bad2);
    printf("V3 (production generated asm) : %.1f G mul/s   mismatches %zu\n", muls / t3 / 1e9, bad3);
    printf("V4 (production, call)         : %.1f G mul/s   mismatches %zu\n", muls / t4 / 1e9, bad4);
    // occupancy sweep: dynamic LDS limits workgroups per CU (256 threads = 1 wave per SIMD each)
    CK(hipFuncSetAttribute((const void *)bench2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)bench2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    Fq *e0, *e1; CK(hipMalloc(&e0, n * sizeof(Fq))); CK(hipMalloc(&e1, n * sizeof(Fq)));
    for (int wps : {1, 2, 3, 4, 8}) {
        const size_t lds = (size_t)(160 * 1024 / wps) & ~(size_t)255;
        auto run = [&](auto kern, Fq *dst) {
            hipEvent_t a0, a1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, (const Fq *)din, dst, iters / 2); CK(hipDeviceSynchronize());
            CK(hipEventRecord(a0));
            for (int r = 0; r < 3; r++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, (const Fq *)din, dst, iters / 2);
            CK(hipEventRecord(a1)); CK(hipEventSynchronize(a1));
            float ms; CK(hipEventElapsedTime(&ms, a0, a1)); return ms / 3 * 1e-3;
        };
        const double ts = run(bench2_kernel<false>, e0), td = run(bench2_kernel<true>, e1);
        std::vector<Fq> q0(n), q1(n);
        CK(hipMemcpy(q0.data(), e0, n * sizeof(Fq), hipMemcpyDeviceToHost)); CK(hipMemcpy(q1.data(), e1, n * sizeof(Fq), hipMemcpyDeviceToHost));
        size_t bad = 0; for (size_t i = 0; i < n; i++) bad += !(q0[i] == q1[i]);
        const double mm = (double)n * (iters / 2) * 4;
        printf("waves/SIMD <= %d: single chain %.1f G mul/s, dual chain (mul2) %.1f G mul/s, mismatches %zu\n", wps, mm / ts / 1e9, mm / td / 1e9, bad);
    }
    {   // sustained rate: ~2 s of back-to-back work (DVFS: the chip lowers its clock under sustained load)
        const int long_iters = 4000;
        double tl = time_kernel(bench_kernel<V3>, dim3(blocks), dim3(threads), 5, (const Fq *)din, d3, long_iters);
        printf("V3 sustained (%.2f s per launch, 6 launches back to back): %.1f G mul/s\n", tl, (double)n * long_iters * 2 / tl / 1e9);
    }
    uint32_t *du; CK(hipMalloc(&du, n * 8));
    const int it2 = 400; const double ops = (double)n * it2 * 64;
    printf("rate v_mad_u64_u32 : %.2f T lane-ops/s\n", ops / time_kernel(rate_mad, dim3(blocks), dim3(threads), 3, du, it2) / 1e12);
    printf("rate v_add_u32     : %.2f T lane-ops/s\n", ops / time_kernel(rate_add, dim3(blocks), dim3(threads), 3, du, it2) / 1e12);
    printf("rate v_addc_co_u32 : %.2f T lane-ops/s\n", ops / time_kernel(rate_addc, dim3(blocks), dim3(threads), 3, du, it2) / 1e12);
    printf("rate v_mul_lo_u32  : %.2f T lane-ops/s\n", ops / time_kernel(rate_mullo, dim3(blocks), dim3(threads), 3, du, it2) / 1e12);
    printf("rate v_mad_u32_u24 : %.2f T lane-ops/s\n", ops / time_kernel(rate_mad24, dim3(blocks), dim3(threads), 3, du, it2) / 1e12);
    printf("rate v_fma_f64     : %.2f T lane-ops/s\n", ops / time_kernel(rate_dfma, dim3(blocks), dim3(threads), 3, (double *)du, it2) / 1e12);
    return 0;
}
