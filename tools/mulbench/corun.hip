// Experiment: how fast does a memory-bound kernel run UNDERNEATH a VALU-saturating kernel on another stream?
// A: field-product loop, 256-lane workgroups, ~110 VGPRs (4 waves per SIMD), grid large enough to occupy the chip for ~40 ms.
// B: streaming kernel (reads u32, LDS-atomic histogram like s2_hist1) with workgroup size WG and U loads in flight per lane.
// Prints B's duration alone and while A is running.
// Build: hipcc -O3 --offload-arch=gfx950 -I../../fawkes-crypto_amd/csrc corun.hip -o corun
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <stdlib.h>
#include "field.hpp"
using namespace fk;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256, 4) void heavy(const Fq *in, Fq *out, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fq a = in[2 * i], b = in[2 * i + 1], c = Fq::add(a, b), d = Fq::sub(a, b), e = Fq::add(c, b), f = Fq::sub(c, a), g = Fq::add(d, a), h = Fq::sub(d, b);
    for (int k = 0; k < iters; k++) {
        Fq x, y;
        Fq::mul2(a, b, c, d, x, y); a = x; c = y;
        Fq::mul2(e, f, g, h, x, y); e = x; g = y;
        Fq::mul2(b, a, d, c, x, y); b = x; d = y;
        Fq::mul2(f, e, h, g, x, y); f = x; h = y;
    }
    out[i] = Fq::add(Fq::add(Fq::add(a, b), Fq::add(c, d)), Fq::add(Fq::add(e, f), Fq::add(g, h)));
}

template <int U>
__global__ void stream_hist(const uint32_t *in, size_t n, uint32_t *out) {
    __shared__ uint32_t hist[512];
    for (uint32_t b = threadIdx.x; b < 512; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    const size_t per = (n + gridDim.x - 1) / gridDim.x, lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (size_t i = lo + threadIdx.x; i < hi; i += (size_t)blockDim.x * U) {
        uint32_t d[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const size_t k = i + (size_t)u * blockDim.x; d[u] = k < hi ? in[k] : 0; }
#pragma unroll
        for (int u = 0; u < U; u++) atomicAdd(&hist[d[u] & 511], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < 512; b += blockDim.x) atomicAdd(&out[b], hist[b]);
}

int main() {
    const size_t n = (size_t)1 << 29;                 // 2 GiB of u32
    uint32_t *din, *dout;
    CK(hipMalloc(&din, n * 4)); CK(hipMalloc(&dout, 4096)); CK(hipMemset(din, 1, n * 4)); CK(hipMemset(dout, 0, 4096));
    const int hb = 256 * 16, ht = 256;
    const size_t hn = (size_t)hb * ht;
    Fq *qin, *qout; CK(hipMalloc(&qin, 2 * hn * sizeof(Fq))); CK(hipMalloc(&qout, hn * sizeof(Fq))); CK(hipMemset(qin, 3, 2 * hn * sizeof(Fq)));
    hipStream_t sa, sb; int plo = 0, phi = 0; CK(hipDeviceGetStreamPriorityRange(&plo, &phi)); printf("priority range: least %d greatest %d, using %s\n", plo, phi, getenv("PRIO") ? "high priority for B" : "equal priorities");
    CK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, plo)); CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, getenv("PRIO") ? phi : plo));
    hipEvent_t e0, e1, a0, a1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1));
    auto heavy_ms = [&](int iters) { hipEventRecord(a0, sa); hipLaunchKernelGGL(heavy, dim3(hb), dim3(ht), 0, sa, qin, qout, iters); hipEventRecord(a1, sa); hipEventSynchronize(a1); float ms; hipEventElapsedTime(&ms, a0, a1); return ms; };
    heavy_ms(50);
    const float t_heavy = heavy_ms(1500);
    printf("heavy kernel alone: %.1f ms\n", t_heavy);
    auto run = [&](auto kern, int wg, int grid, const char *name) {
        auto once = [&]() { hipEventRecord(e0, sb); hipLaunchKernelGGL(kern, dim3(grid), dim3(wg), 0, sb, din, n, dout); hipEventRecord(e1, sb); };
        once(); hipEventSynchronize(e1);
        once(); hipEventSynchronize(e1); float alone; hipEventElapsedTime(&alone, e0, e1);
        hipEventRecord(a0, sa); hipLaunchKernelGGL(heavy, dim3(hb), dim3(ht), 0, sa, qin, qout, 1500); hipEventRecord(a1, sa);
        once(); hipEventSynchronize(e1); hipEventSynchronize(a1);
        float under, hv; hipEventElapsedTime(&under, e0, e1); hipEventElapsedTime(&hv, a0, a1);
        printf("%-28s wg %4d grid %5d: alone %6.2f ms (%.0f GB/s), under the heavy kernel %6.2f ms (heavy took %.1f ms instead of %.1f)\n", name, wg, grid, alone, n * 4 / alone / 1e6, under, hv, t_heavy);
    };
    run(stream_hist<1>, 1024, 1024, "1 load/lane");
    run(stream_hist<8>, 1024, 1024, "8 loads/lane");
    run(stream_hist<1>, 256, 4096, "1 load/lane");
    run(stream_hist<8>, 256, 4096, "8 loads/lane");
    run(stream_hist<16>, 256, 2048, "16 loads/lane");
    run(stream_hist<8>, 64, 16384, "8 loads/lane");
    {   // two streaming kernels on two streams underneath the heavy kernel: fixed pool of leftover capacity, or per-queue trickle?
        hipStream_t sc; CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
        hipEvent_t c0, c1; CK(hipEventCreate(&c0)); CK(hipEventCreate(&c1));
        uint32_t *din2; CK(hipMalloc(&din2, n * 4)); CK(hipMemset(din2, 2, n * 4));
        for (int wg : {64, 256, 1024}) {
            const int grid = (int)(1024 * 1024 / wg);
            hipEventRecord(a0, sa); hipLaunchKernelGGL(heavy, dim3(hb), dim3(ht), 0, sa, qin, qout, 1500); hipEventRecord(a1, sa);
            hipEventRecord(e0, sb); hipLaunchKernelGGL(stream_hist<8>, dim3(grid), dim3(wg), 0, sb, din, n, dout); hipEventRecord(e1, sb);
            hipEventRecord(c0, sc); hipLaunchKernelGGL(stream_hist<8>, dim3(grid), dim3(wg), 0, sc, din2, n, dout); hipEventRecord(c1, sc);
            hipEventSynchronize(e1); hipEventSynchronize(c1); hipEventSynchronize(a1);
            float t1, t2, hv; hipEventElapsedTime(&t1, e0, e1); hipEventElapsedTime(&t2, c0, c1); hipEventElapsedTime(&hv, a0, a1);
            printf("two streaming kernels (wg %d) under the heavy kernel: %.2f ms and %.2f ms (heavy %.1f ms)\n", wg, t1, t2, hv);
        }
    }
    return 0;
}
