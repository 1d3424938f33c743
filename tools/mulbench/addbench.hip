// Microbenchmark: XYZZ mixed additions per second in registers -- the inner loop of the G1 bucket accumulation without its
// memory traffic -- for the 8 x 32-bit accumulator (curve.hpp) and the 9 x 29-bit one (field29.hpp; -DFK_L29_SEQUENTIAL for
// the variant that issues its products one after the other), at 3 and 4 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 -I../../fawkes-crypto_amd/csrc [-DFK_L29_SEQUENTIAL] addbench.hip -o addbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "field29.hpp"
using namespace fk;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MINW>
__global__ __launch_bounds__(256, MINW) void bench32(const G1Affine *pts, G1Xyzz *out, int iters) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    G1Affine q0 = pts[(i * 4) & 1023], q1 = pts[(i * 4 + 1) & 1023], q2 = pts[(i * 4 + 2) & 1023], q3 = pts[(i * 4 + 3) & 1023];
    G1Xyzz acc = G1Xyzz::inf();
    for (int k = 0; k < iters; k++) { acc.add_mixed(q0); acc.add_mixed(q1); acc.add_mixed(affine_neg_if(q2, true)); acc.add_mixed(q3); }
    out[i] = acc;
}
template <int MINW>
__global__ __launch_bounds__(256, MINW) void bench29(const G1Affine *pts, G1Xyzz *out, int iters) {
#if defined(__HIP_DEVICE_COMPILE__)       // field29.hpp is device code
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    G1Affine q0 = pts[(i * 4) & 1023], q1 = pts[(i * 4 + 1) & 1023], q2 = pts[(i * 4 + 2) & 1023], q3 = pts[(i * 4 + 3) & 1023];
    Xyzz29 acc = Xyzz29::inf();
    for (int k = 0; k < iters; k++) { acc.add_mixed(q0, false); acc.add_mixed(q1, false); acc.add_mixed(q2, true); acc.add_mixed(q3, false); }
    out[i] = acc.to_resident();
#endif
}

int main() {
    const int blocks = 256 * 8, threads = 256, iters = 64;
    const size_t n = (size_t)blocks * threads;
    // 1024 distinct points k * G (host arithmetic of curve.hpp)
    std::vector<G1Affine> h(1024);
    G1Affine g; g.x = Fq::from_u64(1); g.y = Fq::from_u64(2);
    G1Xyzz cur = G1Xyzz::from_affine(g);
    for (int i = 0; i < 1024; i++) { h[i] = cur.to_affine(); cur.add_mixed(g); if (i % 7 == 3) cur = G1Xyzz::dbl(cur); }
    G1Affine *dp; G1Xyzz *d32, *d29;
    CK(hipMalloc(&dp, 1024 * sizeof(G1Affine))); CK(hipMalloc(&d32, n * sizeof(G1Xyzz))); CK(hipMalloc(&d29, n * sizeof(G1Xyzz)));
    CK(hipMemcpy(dp, h.data(), 1024 * sizeof(G1Affine), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto launch) { launch(); hipDeviceSynchronize(); hipEventRecord(e0); for (int r = 0; r < 3; r++) launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 3 * 1e-3; };
    const double adds = (double)n * iters * 4;
    const double t32 = timeit([&] { hipLaunchKernelGGL(bench32<4>, dim3(blocks), dim3(threads), 0, 0, dp, d32, iters); });
    const double t29_3 = timeit([&] { hipLaunchKernelGGL(bench29<3>, dim3(blocks), dim3(threads), 0, 0, dp, d29, iters); });
    printf("8 x 32, 4 waves/SIMD : %.2f G mixed additions/s\n", adds / t32 / 1e9);
    printf("9 x 29, 3 waves/SIMD : %.2f G mixed additions/s\n", adds / t29_3 / 1e9);
    std::vector<G1Xyzz> a(n), b(n);
    CK(hipMemcpy(a.data(), d32, n * sizeof(G1Xyzz), hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), d29, n * sizeof(G1Xyzz), hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < n; i += 977) { G1Affine x = a[i].to_affine(), y = b[i].to_affine(); bad += !(x.x == y.x && x.y == y.y); }
    printf("same points: %s\n", bad ? "NO" : "yes");
    const double t29_4 = timeit([&] { hipLaunchKernelGGL(bench29<4>, dim3(blocks), dim3(threads), 0, 0, dp, d29, iters); });
    const double t29_2 = timeit([&] { hipLaunchKernelGGL(bench29<2>, dim3(blocks), dim3(threads), 0, 0, dp, d29, iters); });
    const double t32_3 = timeit([&] { hipLaunchKernelGGL(bench32<3>, dim3(blocks), dim3(threads), 0, 0, dp, d32, iters); });
    printf("9 x 29, 4 waves/SIMD : %.2f\n9 x 29, 2 waves/SIMD : %.2f\n8 x 32, 3 waves/SIMD : %.2f\n", adds / t29_4 / 1e9, adds / t29_2 / 1e9, adds / t32_3 / 1e9);
    return 0;
}
