// Which compute units does a CU-masked HIP stream run on (gfx950, 8 XCDs x 32 CUs)?  For a handful of masks: launch many small workgroups on a stream
// created with hipExtStreamCreateWithCUMask and record (XCC_ID, SE_ID, CU_ID) of every workgroup (s_getreg HW_ID / XCC_ID); print how many distinct
// units each mask reached, per XCD.  Build: hipcc --offload-arch=gfx950 -O2 -o cumask cumask.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <set>
#include <map>

__global__ void where_kernel(uint32_t *out, int spin) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the unit busy for a while so that the launch spreads over everything the mask allows
    volatile uint32_t x = threadIdx.x;
    for (int i = 0; i < spin; i++) x = x * 1664525u + 1013904223u;
    if (threadIdx.x == 0) out[blockIdx.x] = (hw & 0xffffffu) | ((xcc & 0xfu) << 28) | (x & 0);
}

static void run(const char *name, const std::vector<uint32_t> &mask, int ncu) {
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%-34s stream creation failed: %s\n", name, hipGetErrorString(e)); return; }
    const int nb = 16384;
    uint32_t *d; hipMalloc(&d, nb * 4);
    hipLaunchKernelGGL(where_kernel, dim3(nb), dim3(256), 0, st, d, 20000);
    hipStreamSynchronize(st);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, st);
    hipLaunchKernelGGL(where_kernel, dim3(nb), dim3(256), 0, st, d, 20000);
    hipEventRecord(b, st);
    hipStreamSynchronize(st);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    std::vector<uint32_t> h(nb);
    hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost);
    std::map<uint32_t, std::set<uint32_t>> per_xcc;
    for (uint32_t v : h) {
        const uint32_t xcc = v >> 28, cu = (v >> 8) & 0xf, sh = (v >> 12) & 0x1, se = (v >> 13) & 0x7;     // HW_ID: wave 3:0, simd 5:4, (pipe), cu 11:8, sh 12, se 15:13
        per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
    }
    int bits = 0; for (uint32_t w : mask) bits += __builtin_popcount(w);
    size_t total = 0;
    printf("%-34s bits set %3d of %d  time %7.2f ms  units reached per XCD:", name, bits, ncu, ms);
    for (auto &kv : per_xcc) { printf(" x%u:%zu", kv.first, kv.second.size()); total += kv.second.size(); }
    printf("  total %zu\n", total);
    hipFree(d); hipStreamDestroy(st);
}

int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int ncu = pr.multiProcessorCount, words = (ncu + 31) / 32;
    printf("device: %s, %d compute units, mask words %d\n", pr.name, ncu, words);
    auto mk = [&](auto pred) { std::vector<uint32_t> m(words, 0); for (int i = 0; i < ncu; i++) if (pred(i)) m[i / 32] |= 1u << (i % 32); return m; };
    run("all bits", mk([](int) { return true; }), ncu);
    run("bits 0..31 only", mk([](int i) { return i < 32; }), ncu);
    run("bits 0..127 only", mk([](int i) { return i < 128; }), ncu);
    run("bits with i % 8 == 0", mk([](int i) { return i % 8 == 0; }), ncu);
    run("bits with i % 8 != 0", mk([](int i) { return i % 8 != 0; }), ncu);
    run("bits with i % 8 >= 2", mk([](int i) { return i % 8 >= 2; }), ncu);
    run("bits with (i / 8) % 8 == 0", mk([](int i) { return (i / 8) % 8 == 0; }), ncu);
    run("bits with (i / 8) % 8 != 0", mk([](int i) { return (i / 8) % 8 != 0; }), ncu);
    run("bits with i % 2 == 0", mk([](int i) { return i % 2 == 0; }), ncu);
    run("bits with (i / 32) == 0 or 1", mk([](int i) { return i / 32 < 2; }), ncu);
    run("bits with (i / 32) >= 1", mk([](int i) { return i / 32 >= 1; }), ncu);
    run("single bit 0", mk([](int i) { return i == 0; }), ncu);
    run("single bit 1", mk([](int i) { return i == 1; }), ncu);
    run("single bit 8", mk([](int i) { return i == 8; }), ncu);
    return 0;
}
