// Microbenchmark: Montgomery product on 9 x 29-bit limbs (R = 2^261) against the production 8 x 32-bit product.
// With 29-bit limbs a column of up to 18 products (< 2^58 each) plus the carry-in fits a 64-bit accumulator, so
// v_mad_u64_u32 needs no carry-out / v_addc at all and there is no SGPR-carry hazard to schedule around; the price is
// 162 instead of 128 multiply-accumulates and one more limb of state.  Inputs < 2p give outputs < 2p without any
// conditional subtraction (4p^2 / 2^261 + p < 1.01 p).
// Build: hipcc -O3 --offload-arch=gfx950 -I../../fawkes-crypto_amd/csrc limb29.hip -o limb29
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "field.hpp"
using namespace fk;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct L29 { uint32_t v[9]; };
static constexpr uint32_t M29 = (1u << 29) - 1;

struct Consts { uint32_t p[9]; uint32_t inv; };     // p in 29-bit limbs, inv = -p^-1 mod 2^29

static __host__ __device__ inline L29 mul29(const L29 &a, const L29 &b, const Consts &c) {
    uint64_t acc = 0;
    uint32_t m[9];
    L29 r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * c.p[k - i];
        m[k] = ((uint32_t)acc * c.inv) & M29;
        acc += (uint64_t)m[k] * c.p[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) { acc += (uint64_t)a.v[i] * b.v[k - i]; acc += (uint64_t)m[i] * c.p[k - i]; }
        r.v[k - 9] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
    return r;
}

__global__ __launch_bounds__(256) void bench29(const L29 *in, L29 *out, Consts c, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    L29 a = in[2 * i], b = in[2 * i + 1];
    for (int k = 0; k < iters; k++) { a = mul29(a, b, c); b = mul29(b, a, c); }
    L29 o; for (int j = 0; j < 9; j++) o.v[j] = a.v[j] ^ b.v[j];
    out[i] = o;
}
// two independent chains per lane (what mul2 gives the production product)
__global__ __launch_bounds__(256) void bench29x2(const L29 *in, L29 *out, Consts c, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    L29 a = in[2 * i], b = in[2 * i + 1], e = b, f = a;
    e.v[0] ^= 1; f.v[1] ^= 1;
    for (int k = 0; k < iters; k++) { a = mul29(a, b, c); e = mul29(e, f, c); b = mul29(b, a, c); f = mul29(f, e, c); }
    L29 o; for (int j = 0; j < 9; j++) o.v[j] = a.v[j] ^ b.v[j] ^ e.v[j] ^ f.v[j];
    out[i] = o;
}
__global__ __launch_bounds__(256) void bench32(const Fq *in, Fq *out, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fq a = in[2 * i], b = in[2 * i + 1], c = Fq::add(a, b), d = Fq::sub(a, b);
    for (int k = 0; k < iters; k++) { Fq x, y; Fq::mul2(a, b, c, d, x, y); a = x; c = y; Fq::mul2(b, a, d, c, x, y); b = x; d = y; }
    out[i] = Fq::add(Fq::add(a, b), Fq::add(c, d));
}

// ---- host big-int helpers (values as 5 x u64 little endian, enough for 261 + a few bits)
typedef unsigned __int128 u128;
static void to29(const uint32_t w[8], uint32_t o[9]) {      // repack 256 bits
    for (int k = 0; k < 9; k++) {
        uint32_t v = 0;
        for (int b = 0; b < 29; b++) { int bit = 29 * k + b; if (bit < 256 && ((w[bit >> 5] >> (bit & 31)) & 1)) v |= 1u << b; }
        o[k] = v;
    }
}
static void from29(const uint32_t l[9], uint64_t out[5]) {   // limbs may carry a few extra bits in l[8]
    for (int i = 0; i < 5; i++) out[i] = 0;
    for (int k = 0; k < 9; k++) {
        int sh = 29 * k; u128 v = (u128)l[k] << (sh & 63); int w = sh >> 6;
        u128 s = (u128)out[w] + (uint64_t)v; out[w] = (uint64_t)s;
        u128 cy = (s >> 64) + (uint64_t)(v >> 64);
        for (int j = w + 1; j < 5 && cy; j++) { u128 t = (u128)out[j] + (uint64_t)cy; out[j] = (uint64_t)t; cy = t >> 64; }
    }
}

int main() {
    Consts c;
    uint32_t pw[8]; for (int i = 0; i < 8; i++) pw[i] = FqParams::p(i);
    to29(pw, c.p);
    uint32_t inv = 1; for (int i = 0; i < 6; i++) inv *= 2 - c.p[0] * inv;          // p^-1 mod 2^32 (Newton)
    c.inv = (0u - inv) & M29;
    const int blocks = 256 * 8, threads = 256, iters = 200;
    const size_t n = (size_t)blocks * threads;
    std::vector<L29> h(2 * n); std::vector<Fq> hq(2 * n);
    uint64_t s = 777;
    for (size_t i = 0; i < 2 * n; i++) {
        for (int k = 0; k < 8; k++) { s = s * 6364136223846793005ull + 1442695040888963407ull; hq[i].v[k] = (uint32_t)(s >> 32); }
        hq[i].v[7] &= 0x0fffffff;
        to29(hq[i].v, h[i].v);
    }
    // correctness of one product on the host: mul29(a, b) * 2^261 == a * b (mod p), checked through the 8x32 host field:
    // mont32(a, b) = a b 2^-256, so mul29(a,b) == mont32(a,b) * 2^-5  <=>  32 * mul29(a,b) == mont32(a,b) (mod p)
    {
        size_t bad = 0;
        for (size_t i = 0; i < 2000; i++) {
            L29 r = mul29(h[2 * i], h[2 * i + 1], c);
            uint64_t rv[5]; from29(r.v, rv);
            // reduce rv mod p by repeated subtraction (rv < 2p), then times 32 via five doublings in Fq (canonical ints as field elements)
            Fq x; for (int k = 0; k < 8; k++) x.v[k] = (uint32_t)(rv[k >> 1] >> (32 * (k & 1)));
            if (rv[4]) { bad++; continue; }
            x = Fq::reduce_once(x);
            for (int d = 0; d < 5; d++) x = Fq::dbl(x);
            Fq want = Fq::mul(hq[2 * i], hq[2 * i + 1]);
            bad += !(x == want);
        }
        printf("host check of the 9x29 product against the 8x32 product: %zu mismatches of 2000\n", bad);
    }
    L29 *din, *dout; Fq *qin, *qout;
    CK(hipMalloc(&din, 2 * n * sizeof(L29))); CK(hipMalloc(&dout, n * sizeof(L29)));
    CK(hipMalloc(&qin, 2 * n * sizeof(Fq))); CK(hipMalloc(&qout, n * sizeof(Fq)));
    CK(hipMemcpy(din, h.data(), 2 * n * sizeof(L29), hipMemcpyHostToDevice));
    CK(hipMemcpy(qin, hq.data(), 2 * n * sizeof(Fq), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto launch) { launch(); hipDeviceSynchronize(); hipEventRecord(e0); for (int r = 0; r < 3; r++) launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 3 * 1e-3; };
    const double t29 = timeit([&] { hipLaunchKernelGGL(bench29, dim3(blocks), dim3(threads), 0, 0, din, dout, c, iters); });
    const double t29b = timeit([&] { hipLaunchKernelGGL(bench29x2, dim3(blocks), dim3(threads), 0, 0, din, dout, c, iters); });
    const double t32 = timeit([&] { hipLaunchKernelGGL(bench32, dim3(blocks), dim3(threads), 0, 0, qin, qout, iters); });
    // device == host for the 9x29 chain
    std::vector<L29> o(n); CK(hipMemcpy(o.data(), dout, n * sizeof(L29), hipMemcpyDeviceToHost));
    printf("9 x 29-bit limbs, one chain  : %.1f G mul/s\n", (double)n * iters * 2 / t29 / 1e9);
    printf("9 x 29-bit limbs, two chains : %.1f G mul/s\n", (double)n * iters * 4 / t29b / 1e9);
    printf("8 x 32-bit production mul2   : %.1f G mul/s\n", (double)n * iters * 4 / t32 / 1e9);
    return 0;
}
