// Calibration of rocprofv3's FETCH_SIZE for THIS access pattern (MI355X_MICROARCH.md: "other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern"): every lane reads whole 64-byte points (4 x 16-byte loads,
// 64-byte aligned -- how msm_accumulate gathers G1 bases) at pseudo-random indices of a 4 GiB table, so nothing is reused
// and the demand is exactly n * 64 bytes.  Run under `rocprofv3 --pmc FETCH_SIZE`; the factor demand / (FETCH_SIZE * 1024)
// is what tools/pmc_summary.py applies to the accumulate kernels' raw counter.  A second kernel does the same with 128-byte
// points (G2), a third streams the table (the guide's x2 case) as a cross-check.
// Build: hipcc -O3 --offload-arch=gfx950 gathercal.hip -o gathercal
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int W16>   // point width in 16-byte words: 4 = G1, 8 = G2
__global__ __launch_bounds__(256) void gather_kernel(const uint4 *tab, uint64_t npoints, uint32_t per_lane, uint4 *out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    uint64_t s = t * 0x9E3779B97F4A7C15ull + 12345;
    for (uint32_t k = 0; k < per_lane; k++) {
        s ^= s >> 29; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 32;
        const uint4 *p = tab + (s % npoints) * W16;
#pragma unroll
        for (int j = 0; j < W16; j++) { const uint4 v = p[j]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    }
    out[t] = acc;
}
__global__ __launch_bounds__(256) void stream_kernel(const uint4 *tab, uint64_t nwords, uint4 *out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (uint64_t i = t; i < nwords; i += stride) { const uint4 v = tab[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    out[t] = acc;
}

int main() {
    const uint64_t bytes = (uint64_t)4 << 30;
    uint4 *tab, *out;
    CK(hipMalloc(&tab, bytes)); CK(hipMemset(tab, 0x5a, bytes));
    const unsigned blocks = 256 * 16, threads = 256; const uint64_t lanes = (uint64_t)blocks * threads; const uint32_t per = 64;
    CK(hipMalloc(&out, lanes * sizeof(uint4)));
    hipLaunchKernelGGL(gather_kernel<4>, dim3(blocks), dim3(threads), 0, 0, tab, bytes / 64, per, out);
    hipLaunchKernelGGL(gather_kernel<8>, dim3(blocks), dim3(threads), 0, 0, tab, bytes / 128, per, out);
    hipLaunchKernelGGL(stream_kernel, dim3(blocks), dim3(threads), 0, 0, tab, bytes / 16, out);
    CK(hipDeviceSynchronize());
    printf("gather_kernel<4>: demand %llu bytes\ngather_kernel<8>: demand %llu bytes\nstream_kernel: demand %llu bytes\n",
           (unsigned long long)(lanes * per * 64), (unsigned long long)(lanes * per * 128), (unsigned long long)bytes);
    return 0;
}
