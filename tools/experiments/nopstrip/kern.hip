#include <hip/hip_runtime.h>
#include "curve.hpp"
using namespace fk;
extern "C" __global__ __launch_bounds__(256, 4) void bench_xyzz(const G1Affine *pts, G1Xyzz *out, int iters) {
    using FL = FqL;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Affine<FL> q[4];
    for (int j = 0; j < 4; j++) { G1Affine p = pts[(i * 4 + j) & 1023]; __builtin_memcpy(&q[j], &p, sizeof p); }
    Xyzz<FL> acc = Xyzz<FL>::inf();
    for (int k = 0; k < iters; k++) { acc.add_mixed(q[0]); acc.add_mixed_nz(q[1]); acc.add_mixed_nz(q[2]); acc.add_mixed_nz(q[3]); }
    out[i] = acc.is_inf() ? G1Xyzz::inf() : G1Xyzz{canon(acc.x), canon(acc.y), canon(acc.zz), canon(acc.zzz)};
}
