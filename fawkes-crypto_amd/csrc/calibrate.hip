// Live calibration of the two ceilings the measurement (bench.py) quotes for the VALU-bound kernels: the issue rate of
// v_mad_u64_u32 -- the 32 x 32 -> 64-bit multiply-accumulate every 256-bit Montgomery product is made of -- and the rate
// of the production Montgomery product (field.hpp: Fq::mul2, two interleaved chains) running alone in registers.
// Measured on the device the benchmark runs on, so that the roofline fractions do not depend on a clock assumption.
#include "common.hpp"

namespace fk {

__global__ __launch_bounds__(256) void calib_mad_kernel(uint32_t *out, int iters) {
    uint32_t x = threadIdx.x + 1, y = blockIdx.x + 3;
    uint64_t a0 = x, a1 = y, a2 = x ^ y, a3 = x + y;
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y) : "vcc");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 + a1 + a2 + a3);
}

__global__ __launch_bounds__(256) void calib_mul_kernel(Fq *out, int iters) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fq a = Fq::one(), b = Fq::r2();
    a.v[0] ^= (uint32_t)i; b.v[1] ^= (uint32_t)(i * 2654435761u); a.v[7] &= 0x0fffffffu; b.v[7] &= 0x0fffffffu;
    Fq c = Fq::add(a, b), d = Fq::sub(a, b);
    for (int k = 0; k < iters; k++) {
        Fq x, y;
        Fq::mul2(a, b, c, d, x, y); a = x; c = y;
        Fq::mul2(b, a, d, c, x, y); b = x; d = y;
    }
    out[i] = Fq::add(Fq::add(a, b), Fq::add(c, d));
}

}  // namespace fk

using namespace fk;

extern "C" int fk_calibrate(fk_ctx *ctx, double out[2]) { return fk_guard(ctx, [&]() -> int {
    if (!ctx || !out) return FK_ERR_BAD_ARG;
    FK_HIP(ctx, hipSetDevice(ctx->device));
    const unsigned blocks = 256 * 8, threads = 256;
    const size_t n = (size_t)blocks * threads;
    FK_HIP(ctx, ctx->misc.reserve(n * sizeof(Fq)));
    hipEvent_t e0, e1;
    FK_HIP(ctx, hipEventCreate(&e0)); FK_HIP(ctx, hipEventCreate(&e1));
    auto timed = [&](auto launch, double *sec) -> int {
        launch();                                     // warm-up (code load, clocks)
        FK_HIP(ctx, hipEventRecord(e0, ctx->stream));
        for (int r = 0; r < 3; r++) launch();
        FK_HIP(ctx, hipEventRecord(e1, ctx->stream));
        FK_HIP(ctx, hipEventSynchronize(e1));
        FK_HIP(ctx, hipGetLastError());
        float ms = 0; FK_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
        *sec = ms / 3 * 1e-3;
        return FK_OK;
    };
    double t_mad = 0, t_mul = 0;
    const int it_mad = 400, it_mul = 100;
    int rc = timed([&] { hipLaunchKernelGGL(calib_mad_kernel, dim3(blocks), dim3(threads), 0, ctx->stream, ctx->misc.as<uint32_t>(), it_mad); }, &t_mad);
    if (rc == FK_OK) rc = timed([&] { hipLaunchKernelGGL(calib_mul_kernel, dim3(blocks), dim3(threads), 0, ctx->stream, ctx->misc.as<Fq>(), it_mul); }, &t_mul);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (rc != FK_OK) return rc;
    out[0] = (double)n * it_mad * 64 / t_mad;         // v_mad_u64_u32 lane-operations per second
    out[1] = (double)n * it_mul * 4 / t_mul;          // Montgomery products per second, production multiplier in isolation
    return FK_OK;
}); }
