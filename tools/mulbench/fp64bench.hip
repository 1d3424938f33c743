// Bounded experiment: a Montgomery product on the FP64 FMA pipe -- 5 x 52-bit limbs held as doubles, the high and
// low half of every partial product by two v_fma_f64 under round-toward-zero (MODE register, s_setreg: not a per-instruction modifier on
// CDNA), column sums as 64-bit integer adds on the doubles' bit patterns (Emmart / Zheng / Weems' "DPF" form).
//
// KILL CRITERION, written before the first run: keep the idea only if the complete product can reach >= 1.15 x the shipped
// 126 G products/s.  A Montgomery product is two 5 x 5 limb products (a * b, then q * p) plus the q computation and the carry
// resolution, so ITS a * b HALF ALONE -- 25 partial products, measured here in registers with nothing else around it -- must run at
// >= 2 x 1.15 x 126 = 290 G half-products/s for the whole to have a chance.  Below that the idea is dead without building the rest.
//
// Also printed: the issue rates of the instructions the form is made of (v_fma_f64, v_add_f64, v_lshl_add_u64, the 32-bit add pair),
// so that the instruction count that decides the matter can be priced.
// Build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 fp64bench.hip -o fp64bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

static __device__ __forceinline__ void set_dp_round_toward_zero() {
    // MODE[3:2] = FP_ROUND for f64 / f16: 3 = toward zero
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");
}

constexpr double C1 = 0x1p104;                 // hi = fma(a, b, 2^104) truncated: 2^104 + floor(ab / 2^52) * 2^52
constexpr double C2 = 0x1p104 + 0x1p52;        // lo = fma(a, b, C2 - hi) = (ab mod 2^52) + 2^52: its mantissa IS the low half

// one partial product into two 64-bit column accumulators: 2 x v_fma_f64 + 1 x v_add_f64 + 2 x 64-bit integer add
static __device__ __forceinline__ void pp(double a, double b, uint64_t &col_lo, uint64_t &col_hi) {
    const double hi = __builtin_fma(a, b, C1);
    const double sub = C2 - hi;
    const double lo = __builtin_fma(a, b, sub);
    col_lo += (uint64_t)__double_as_longlong(lo);
    col_hi += (uint64_t)__double_as_longlong(hi);
}

// correctness: the ten column sums of one 5 x 5 product (exponent patterns removed)
__global__ void dpf_check_kernel(const double *a, const double *b, uint64_t *out) {
    set_dp_round_toward_zero();
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double x[5], y[5];
    for (int i = 0; i < 5; i++) { x[i] = a[t * 5 + i]; y[i] = b[t * 5 + i]; }
    uint64_t acc[11] = {0};
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int j = 0; j < 5; j++) pp(x[i], y[j], acc[i + j], acc[i + j + 1]);
    // column k received n_lo(k) low halves (exponent pattern of 2^52) and n_hi(k) high halves (pattern of 2^104)
    for (int k = 0; k < 10; k++) {
        const int n_lo = (k <= 4 ? k + 1 : 9 - k), n_hi = (k >= 1 ? (k - 1 <= 4 ? k : 10 - k) : 0);
        acc[k] -= (uint64_t)(k <= 8 ? n_lo : 0) * 0x4330000000000000ull + (uint64_t)n_hi * 0x4670000000000000ull;
        out[t * 10 + k] = acc[k];
    }
}

// throughput: IT dependent half-products per lane, four independent chains per lane (the mixed addition has that much parallelism)
template <int CH>
__global__ __launch_bounds__(256) void dpf_rate_kernel(double *io, int iters) {
    set_dp_round_toward_zero();
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double x[CH][5], y[5];
    for (int c = 0; c < CH; c++) for (int i = 0; i < 5; i++) x[c][i] = io[(t * CH + c) * 5 + i];
    for (int i = 0; i < 5; i++) y[i] = io[i] + 3.0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            uint64_t acc[11] = {0};
#pragma unroll
            for (int i = 0; i < 5; i++)
#pragma unroll
                for (int j = 0; j < 5; j++) pp(x[c][i], y[j], acc[i + j], acc[i + j + 1]);
            // fold the ten columns back into five 52-bit limbs held as doubles (so that the next product depends on this one): the
            // cheapest possible stand-in for "the rest of the product" -- 5 x (xor, and, or) on 64 bits + 5 v_add_f64
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const uint64_t v = ((acc[i] ^ acc[i + 5]) & 0x000fffffffffffffull) | 0x4330000000000000ull;
                x[c][i] = __longlong_as_double((long long)v) - 0x1p52;
            }
        }
    }
    for (int c = 0; c < CH; c++) for (int i = 0; i < 5; i++) io[(t * CH + c) * 5 + i] = x[c][i];
}

// ---- raw issue rates
__global__ __launch_bounds__(256) void rate_fma_kernel(double *io, int iters) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double a[8]; for (int i = 0; i < 8; i++) a[i] = io[t * 8 + i];
    const double m = io[0] + 1.0, c = io[1];
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
    for (int i = 0; i < 8; i++) io[t * 8 + i] = a[i];
}
__global__ __launch_bounds__(256) void rate_dadd_kernel(double *io, int iters) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double a[8]; for (int i = 0; i < 8; i++) a[i] = io[t * 8 + i];
    const double c = io[1];
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
    for (int i = 0; i < 8; i++) io[t * 8 + i] = a[i];
}
__global__ __launch_bounds__(256) void rate_lshl_add_u64_kernel(uint64_t *io, int iters) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t a[8]; for (int i = 0; i < 8; i++) a[i] = io[t * 8 + i];
    const uint64_t c = io[1];
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "v"(c));
    for (int i = 0; i < 8; i++) io[t * 8 + i] = a[i];
}
__global__ __launch_bounds__(256) void rate_add_pair_kernel(uint64_t *io, int iters) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t a[8]; for (int i = 0; i < 8; i++) a[i] = io[t * 8 + i];
    const uint64_t c = io[1];
    const uint32_t cl = (uint32_t)c, ch = (uint32_t)(c >> 32);
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint32_t lo = (uint32_t)a[i], hi = (uint32_t)(a[i] >> 32);
            asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(cl), "v"(ch) : "vcc");
            a[i] = lo | ((uint64_t)hi << 32);
        }
    for (int i = 0; i < 8; i++) io[t * 8 + i] = a[i];
}

int main() {
    const unsigned blocks = 256 * 8, threads = 256;
    const size_t n = (size_t)blocks * threads;
    // ---- correctness of the hi / lo split under the MODE register's rounding
    {
        const size_t nt = 4096;
        std::vector<double> a(nt * 5), b(nt * 5);
        std::vector<uint64_t> ai(nt * 5), bi(nt * 5);
        uint64_t s = 0x9E3779B97F4A7C15ull;
        auto next = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
        for (size_t i = 0; i < nt * 5; i++) {
            ai[i] = next() & 0x000fffffffffffffull; bi[i] = next() & 0x000fffffffffffffull;
            if (i % 97 == 0) ai[i] = 0x000fffffffffffffull;
            if (i % 89 == 0) bi[i] = 0x000fffffffffffffull;
            if (i % 101 == 0) ai[i] = 0;
            a[i] = (double)ai[i]; b[i] = (double)bi[i];
        }
        double *da, *db; uint64_t *dout;
        CK(hipMalloc(&da, nt * 5 * 8)); CK(hipMalloc(&db, nt * 5 * 8)); CK(hipMalloc(&dout, nt * 10 * 8));
        CK(hipMemcpy(da, a.data(), nt * 5 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), nt * 5 * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(dpf_check_kernel, dim3(nt / 256), dim3(256), 0, 0, da, db, dout);
        CK(hipDeviceSynchronize());
        std::vector<uint64_t> out(nt * 10);
        CK(hipMemcpy(out.data(), dout, nt * 10 * 8, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t t = 0; t < nt; t++) {
            uint64_t want[11] = {0};
            for (int i = 0; i < 5; i++) for (int j = 0; j < 5; j++) {
                const unsigned __int128 p = (unsigned __int128)ai[t * 5 + i] * bi[t * 5 + j];
                want[i + j] += (uint64_t)(p & 0x000fffffffffffffull); want[i + j + 1] += (uint64_t)(p >> 52);
            }
            for (int k = 0; k < 10; k++) if (out[t * 10 + k] != want[k]) bad++;
        }
        printf("hi / lo split under MODE.FP_ROUND(f64) = toward zero: %zu of %zu column sums wrong\n", bad, nt * 10);
        (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double *dio; CK(hipMalloc(&dio, n * 4 * 5 * 8)); CK(hipMemset(dio, 0, n * 4 * 5 * 8));
    {
        std::vector<double> init(n * 4 * 5);
        for (size_t i = 0; i < init.size(); i++) init[i] = (double)((i * 0x9E3779B97F4A7C15ull) & 0x000fffffffffffffull);
        CK(hipMemcpy(dio, init.data(), init.size() * 8, hipMemcpyHostToDevice));
    }
    auto timed = [&](auto launch, int reps) -> double {
        launch(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        for (int r = 0; r < reps; r++) launch();
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        return ms / reps * 1e-3;
    };
    const int it = 200;
    double t1 = timed([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(dpf_rate_kernel<1>), dim3(blocks), dim3(threads), 0, 0, dio, it); }, 3);
    double t2 = timed([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(dpf_rate_kernel<2>), dim3(blocks), dim3(threads), 0, 0, dio, it); }, 3);
    double t4 = timed([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(dpf_rate_kernel<4>), dim3(blocks), dim3(threads), 0, 0, dio, it); }, 3);
    printf("DPF a*b half (25 partial products: 50 v_fma_f64 + 25 v_add_f64 + 50 64-bit adds, + fold 5 limbs):\n");
    printf("    1 chain / lane : %7.1f G half-products/s\n", (double)n * it * 1 / t1 / 1e9);
    printf("    2 chains / lane: %7.1f G half-products/s\n", (double)n * it * 2 / t2 / 1e9);
    printf("    4 chains / lane: %7.1f G half-products/s\n", (double)n * it * 4 / t4 / 1e9);
    printf("    needed for 1.15 x the shipped product (126 G/s): >= 290 G half-products/s\n");
    const int ir = 400;
    double tf = timed([&] { hipLaunchKernelGGL(rate_fma_kernel, dim3(blocks), dim3(threads), 0, 0, dio, ir); }, 3);
    double ta = timed([&] { hipLaunchKernelGGL(rate_dadd_kernel, dim3(blocks), dim3(threads), 0, 0, dio, ir); }, 3);
    double tl = timed([&] { hipLaunchKernelGGL(rate_lshl_add_u64_kernel, dim3(blocks), dim3(threads), 0, 0, (uint64_t *)dio, ir); }, 3);
    double tp = timed([&] { hipLaunchKernelGGL(rate_add_pair_kernel, dim3(blocks), dim3(threads), 0, 0, (uint64_t *)dio, ir); }, 3);
    printf("rate v_fma_f64                      : %6.2f T lane-ops/s\n", (double)n * ir * 8 / tf / 1e12);
    printf("rate v_add_f64                      : %6.2f T lane-ops/s\n", (double)n * ir * 8 / ta / 1e12);
    printf("rate v_lshl_add_u64                 : %6.2f T lane-ops/s\n", (double)n * ir * 8 / tl / 1e12);
    printf("rate v_add_co_u32 + v_addc_co_u32   : %6.2f T 64-bit adds/s (two instructions each)\n", (double)n * ir * 8 / tp / 1e12);
    return 0;
}
