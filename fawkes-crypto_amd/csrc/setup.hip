// Groth16 key generation on the GPU.
//
// Restates bellman_ce::groth16::generator::generate_parameters, which fawkes reaches
// from /root/reference/fawkes-crypto/src/backend/bellman_groth16/setup.rs:20, for explicit toxic waste
// (tau, alpha, beta, gamma, delta).  It exists so that VALID proving keys can be produced at the sizes the
// benchmark configurations name (2^20 and up), where a CPU setup takes minutes to hours: with a valid key a
// GPU proof can be checked against the Groth16 pairing equation at full size.
//
//   powers of tau            two-level table expansion (one multiply per element)
//   h[i] = g1^(tau^i (tau^m - 1)/delta)
//   Lagrange L_j(tau)        iNTT of the power vector (the shared NTT kernels)
//   A_k, B_k, C_k            transposed sparse product over a CSC copy of the R1CS (one lane per variable)
//   a, b_g1, b_g2, ic, l     fixed-base scalar multiplication: 32 windows x 8 bits, table in L2, XYZZ mixed adds
//   a, b_g1, b_g2            compacted to the non-identity points (inputs first, then aux) -- the layout the
//                            prover's density-filtered queries index into
#include "common.hpp"
#include <string.h>
#include <unordered_map>
#include <string>

namespace fk {

__global__ void expand_powers_kernel(const Fr *lo, const Fr *hi, uint32_t L, size_t n, Fr *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = Fr::mul(lo[i & ((1u << L) - 1)], hi[i >> L]);
}
__global__ void scale_kernel(const Fr *in, Fr c, size_t n, Fr *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = Fr::mul(in[i], c);
}

// out[v] = sum over the column's entries  table[cidx] * lag[row]   (+ lag[num_gates + v] for inputs in matrix 0)
static constexpr uint64_t CSC_HEAVY = 4096;      // columns with more entries go through csc_heavy_kernel
static constexpr uint64_t CSC_SEG = 16384;       // entries per workgroup there
struct CscArgs { const uint64_t *ptr[3]; const uint32_t *row[3]; const uint32_t *cidx[3]; Fr *out[3]; };
__global__ __launch_bounds__(256) void csc_eval_kernel(CscArgs a, const Fr *table, const Fr *lag, uint64_t num_gates, uint32_t num_input, uint64_t nv) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t mtx = blockIdx.y;
    if (v >= nv) return;
    Fr acc = Fr::zero();
    uint64_t k0 = a.ptr[mtx][v], k1 = a.ptr[mtx][v + 1];
    if (k1 - k0 > CSC_HEAVY) k1 = k0;            // heavy column (e.g. the constant ONE): summed by csc_heavy_kernel
    for (uint64_t k = k0, e = k1; k < e; k++) {
        Fr t = lag[a.row[mtx][k]];
        const uint32_t ci = a.cidx[mtx][k];
        if (ci) t = Fr::mul(t, table[ci]);
        acc = Fr::add(acc, t);
    }
    if (mtx == 0 && v < num_input) acc = Fr::add(acc, lag[num_gates + v]);
    a.out[mtx][v] = acc;
}

// Batch circuits (fk_setup_tiled): the CSC describes ONE instance and the system is `copies` of it, copy j's gates being
// rows [j*base_gates, (j+1)*base_gates).  sc[mtx][j*nvb + v] = sum over instance column v of table[cidx] * lag[j*base_gates + row];
// tile_assemble_kernel then lays the values out in the batch's variable order (ONE shared -> the sum over the copies).
__global__ __launch_bounds__(256) void csc_eval_tiled_kernel(CscArgs a, const Fr *table, const Fr *lag, uint32_t base_gates, uint32_t nvb) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x, mtx = blockIdx.y, copy = blockIdx.z;
    if (v >= nvb) return;
    const Fr *lg = lag + (uint64_t)copy * base_gates;
    Fr acc = Fr::zero();
    for (uint64_t k = a.ptr[mtx][v], e = a.ptr[mtx][v + 1]; k < e; k++) {
        Fr t = lg[a.row[mtx][k]];
        const uint32_t ci = a.cidx[mtx][k];
        if (ci) t = Fr::mul(t, table[ci]);
        acc = Fr::add(acc, t);
    }
    a.out[mtx][(uint64_t)copy * nvb + v] = acc;
}
struct TileAsm { const Fr *sc[3]; Fr *out[3]; };
__global__ __launch_bounds__(256) void tile_assemble_kernel(TileAsm a, const Fr *lag, uint32_t base_input, uint32_t base_aux, uint32_t copies, uint64_t num_gates,
                                                            uint32_t num_input, uint64_t nv) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t mtx = blockIdx.y, nvb = base_input + base_aux;
    if (v >= nv) return;
    Fr acc;
    if (v == 0) {
        acc = Fr::zero();
        for (uint32_t j = 0; j < copies; j++) acc = Fr::add(acc, a.sc[mtx][(uint64_t)j * nvb]);
    } else if (v < num_input) {
        const uint32_t j = (uint32_t)(v - 1) / (base_input - 1), i = 1 + (uint32_t)(v - 1) % (base_input - 1);
        acc = a.sc[mtx][(uint64_t)j * nvb + i];
    } else {
        const uint32_t q = (uint32_t)(v - num_input), j = q / base_aux, i = q % base_aux;
        acc = a.sc[mtx][(uint64_t)j * nvb + base_input + i];
    }
    if (mtx == 0 && v < num_input) acc = Fr::add(acc, lag[num_gates + v]);      // bellman's input_i * 0 = 0 rows
    a.out[mtx][v] = acc;
}

// one workgroup per CSC_SEG-entry segment of a heavy column: partial[seg] = sum table[cidx] * lag[row]
struct HeavySeg { uint32_t mtx; uint32_t pad; uint64_t lo, hi; };
__global__ __launch_bounds__(256) void csc_heavy_kernel(CscArgs a, const Fr *table, const Fr *lag, const HeavySeg *segs, Fr *partial) {
    __shared__ Fr sh[256];
    const HeavySeg sg = segs[blockIdx.x];
    Fr acc = Fr::zero();
    for (uint64_t k = sg.lo + threadIdx.x; k < sg.hi; k += 256) {
        Fr t = lag[a.row[sg.mtx][k]];
        const uint32_t ci = a.cidx[sg.mtx][k];
        if (ci) t = Fr::mul(t, table[ci]);
        acc = Fr::add(acc, t);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t off = 128; off; off >>= 1) {
        if (threadIdx.x < off) sh[threadIdx.x] = Fr::add(sh[threadIdx.x], sh[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

// e[v] = (beta A_v + alpha B_v + C_v) * (v < num_input ? 1/gamma : 1/delta); flags: A_v != 0, B_v != 0
__global__ void combine_kernel(const Fr *A, const Fr *B, const Fr *C, Fr beta, Fr alpha, Fr ginv, Fr dinv, uint32_t num_input, uint64_t nv,
                               Fr *e, uint8_t *fa, uint8_t *fb) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    Fr x = Fr::add(Fr::add(Fr::mul(A[v], beta), Fr::mul(B[v], alpha)), C[v]);
    e[v] = Fr::mul(x, v < num_input ? ginv : dinv);
    fa[v] = A[v].is_zero() ? 0 : 1;
    fb[v] = B[v].is_zero() ? 0 : 1;
}

// table[w * 255 + (d - 1)] = d * 2^(8w) * G, affine.  out[i] = scalar[i] * G.
template <class F>
__global__ __launch_bounds__(128) void fixed_base_kernel(const Affine<F> *table, const Fr *scalars, size_t n, Affine<F> *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr k = Fr::from_mont(scalars[i]);
    Xyzz<F> acc = Xyzz<F>::inf();
    for (int w = 0; w < 32; w++) {
        const uint32_t d = (k.v[w >> 2] >> ((w & 3) * 8)) & 0xff;
        if (d) acc.add_mixed(table[w * 255 + d - 1]);
    }
    out[i] = acc.to_affine();
}

// ---- stream compaction of fixed-size elements by a byte flag (order preserving)
static constexpr uint32_t CE_BLOCK = 2048;
__global__ __launch_bounds__(256) void ce_count_kernel(const uint8_t *flag, size_t n, uint32_t *blk) {
    __shared__ uint32_t sh[256];
    const size_t base = (size_t)blockIdx.x * CE_BLOCK + (size_t)threadIdx.x * 8;
    uint32_t c = 0;
    for (int k = 0; k < 8; k++) if (base + k < n && flag[base + k]) c++;
    sh[threadIdx.x] = c;
    __syncthreads();
    for (uint32_t off = 128; off; off >>= 1) { if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off]; __syncthreads(); }
    if (threadIdx.x == 0) blk[blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(1024) void ce_scan_kernel(uint32_t *v, uint32_t n, uint32_t *total) {
    __shared__ uint32_t part[1024];
    const uint32_t tid = threadIdx.x, ipt = (n + 1023) / 1024;
    const uint32_t lo = tid * ipt, hi = lo + ipt < n ? lo + ipt : n;
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi && lo < n; i++) sum += v[i];
    part[tid] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        uint32_t x = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += x;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;
    for (uint32_t i = lo; i < hi && lo < n; i++) { uint32_t x = v[i]; v[i] = run; run += x; }
    if (tid == 1023) *total = part[1023];
}
template <class T>
__global__ __launch_bounds__(256) void ce_scatter_kernel(const T *in, const uint8_t *flag, size_t n, const uint32_t *blk, T *out) {
    __shared__ uint32_t sh[256];
    const size_t base = (size_t)blockIdx.x * CE_BLOCK + (size_t)threadIdx.x * 8;
    uint32_t c = 0;
    for (int k = 0; k < 8; k++) if (base + k < n && flag[base + k]) c++;
    sh[threadIdx.x] = c;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {
        uint32_t x = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += x;
        __syncthreads();
    }
    size_t pos = (size_t)blk[blockIdx.x] + sh[threadIdx.x] - c;
    for (int k = 0; k < 8; k++) if (base + k < n && flag[base + k]) out[pos++] = in[base + k];
}

template <class T>
static int compact_elems(fk_ctx *ctx, const T *d_in, const uint8_t *d_flag, size_t n, T *d_out, uint64_t *n_out) {
    *n_out = 0;
    if (!n) return FK_OK;
    const uint32_t nb = (uint32_t)((n + CE_BLOCK - 1) / CE_BLOCK);
    FK_HIP(ctx, ctx->scan_tmp.reserve((size_t)nb * 4 + 16));
    uint32_t *blk = ctx->scan_tmp.as<uint32_t>(), *d_total = blk + nb;
    hipLaunchKernelGGL(ce_count_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_flag, n, blk);
    hipLaunchKernelGGL(ce_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, blk, nb, d_total);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ce_scatter_kernel<T>), dim3(nb), dim3(256), 0, ctx->stream, d_in, d_flag, n, blk, d_out);
    FK_HIP(ctx, hipGetLastError());
    uint32_t total = 0;
    FK_HIP(ctx, hipMemcpyAsync(&total, d_total, 4, hipMemcpyDeviceToHost, ctx->stream));
    FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = total;
    return FK_OK;
}

// host: fixed-base table d * 2^(8w) * G for w < 32, d in 1..255, affine via one batched inversion
template <class F>
static std::vector<Affine<F>> host_fb_table(const Affine<F> &g) {
    std::vector<Xyzz<F>> pts(32 * 255);
    Xyzz<F> base = Xyzz<F>::from_affine(g);
    for (int w = 0; w < 32; w++) {
        Xyzz<F> cur = base;
        for (int d = 1; d <= 255; d++) { pts[w * 255 + d - 1] = cur; cur.add(base); }
        base = cur;  // 256 * previous base
    }
    // batched XYZZ -> affine: invert prod(zz * zzz)
    const size_t n = pts.size();
    std::vector<F> pre(n);
    F run = F::one();
    for (size_t i = 0; i < n; i++) { pre[i] = run; run = F::mul(run, F::mul(pts[i].zz, pts[i].zzz)); }
    F inv = F::inv(run);
    std::vector<Affine<F>> out(n);
    for (size_t i = n; i-- > 0;) {
        const F t = F::mul(inv, pre[i]);                       // 1 / (zz_i zzz_i)
        inv = F::mul(inv, F::mul(pts[i].zz, pts[i].zzz));
        out[i] = Affine<F>{F::mul(pts[i].x, F::mul(t, pts[i].zzz)), F::mul(pts[i].y, F::mul(t, pts[i].zz))};
    }
    return out;
}

template <class F>
static Affine<F> host_mul(const Affine<F> &p, const Fr &k_mont) {
    const Fr k = Fr::from_mont(k_mont);
    return Xyzz<F>::mul_scalar(Xyzz<F>::from_affine(p), k.v).to_affine();
}

static Fr fr_load(const uint64_t *p) { Fr r; memcpy(&r, p, 32); return r; }

}  // namespace fk

using namespace fk;

struct DevFree { std::vector<void *> v; ~DevFree() { for (void *p : v) if (p) (void)hipFree(p); } };

extern "C" {

// cs: one instance; the key is for `copies` of it as one system (1 = the system itself)
static int setup_impl(fk_ctx *ctx, const fk_r1cs *cs, uint32_t copies, const uint64_t tau_[4], const uint64_t alpha_[4], const uint64_t beta_[4],
                      const uint64_t gamma_[4], const uint64_t delta_[4], uint32_t shard_index, uint32_t shard_count, double z_frac_lo, double z_frac_hi,
                      fk_key **out_key, uint8_t vk_out[6 * 128], uint8_t *ic_out) {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (!cs || !tau_ || !alpha_ || !beta_ || !gamma_ || !delta_ || !out_key || !vk_out || !ic_out) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "setup: null argument");
    *out_key = nullptr;
    if (shard_count == 0 || shard_index >= shard_count) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "setup: bad shard %u/%u", shard_index, shard_count);
    if (z_frac_lo >= 0.0 && (!(z_frac_hi <= 1.0 && z_frac_lo <= z_frac_hi) || (shard_count == 1 && !(z_frac_lo == 0.0 && z_frac_hi >= 1.0))))
        FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "setup: bad z fraction range [%g, %g)", z_frac_lo, z_frac_hi);
    FK_HIP(ctx, hipSetDevice(ctx->device));
    const Fr tau = fr_load(tau_), alpha = fr_load(alpha_), beta = fr_load(beta_), gamma = fr_load(gamma_), delta = fr_load(delta_);
    if (gamma.is_zero() || delta.is_zero()) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "setup: gamma and delta must be non-zero");
    if (cs->num_input == 0) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "setup: num_input must include the constant ONE");
    if (copies == 0) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "setup: copies must be at least 1");
    // totals of the batch: ONE is shared, inputs and aux variables are per copy (spmv.hip: fk_r1cs_load_tiled)
    const uint64_t t_in = 1 + (uint64_t)copies * (cs->num_input - 1), t_aux = (uint64_t)copies * cs->num_aux, t_gates = (uint64_t)copies * cs->num_gates;
    const uint64_t nvb = (uint64_t)cs->num_input + cs->num_aux;     // variables of one instance
    if (t_in + t_aux > 0xffffffffull || t_gates + t_in > 0xffffffffull || (copies > 1 && (cs->num_gates == 0 || copies > 65535)))
        FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "setup: %u copies of this system are out of range", copies);
    const uint32_t num_input = (uint32_t)t_in, num_aux = (uint32_t)t_aux;
    const uint64_t rows = t_gates + t_in;
    const uint32_t log_m = ceil_log2_u64(rows);
    if (log_m >= FK_FR_S) FK_SET_ERR(ctx, FK_ERR_DOMAIN_TOO_LARGE, "setup: evaluation domain 2^%u too large (max 2^%d)", log_m, FK_FR_S - 1);
    const uint64_t m = (uint64_t)1 << log_m;
    const uint64_t nv = t_in + t_aux;
    const uint64_t *ptrs[3] = {cs->a_ptr, cs->b_ptr, cs->c_ptr};
    const uint32_t *cols[3] = {cs->a_col, cs->b_col, cs->c_col};
    const uint64_t *vals[3] = {cs->a_val, cs->b_val, cs->c_val};
    for (int k = 0; k < 3; k++) {
        if (!ptrs[k]) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "setup: null row pointer");
        const uint64_t nnz = ptrs[k][cs->num_gates];
        for (uint64_t i = 0; i < nnz; i++) if (cols[k][i] >= nvb) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "setup: variable index %u out of range", cols[k][i]);
    }
    DevFree tmp;
    auto dalloc = [&](size_t bytes) -> void * { void *p = nullptr; if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr; tmp.v.push_back(p); return p; };
    hipStream_t st = ctx->stream;

    // ---- CSC copies with dictionary-coded coefficients (slot 0 = ONE)
    std::unordered_map<std::string, uint32_t> dict;
    std::vector<Fr> table; const Fr one = Fr::one();
    table.push_back(one); dict.emplace(std::string((const char *)&one, 32), 0u);
    CscArgs ca;
    Fr *d_abc[3];
    Fr last = one; uint32_t last_idx = 0;
    std::vector<HeavySeg> heavy_segs; std::vector<uint64_t> heavy_var;
    for (int k = 0; k < 3; k++) {
        const uint64_t nnz = ptrs[k][cs->num_gates];
        std::vector<uint64_t> cptr(nvb + 1, 0);
        for (uint64_t i = 0; i < nnz; i++) cptr[cols[k][i] + 1]++;
        for (uint64_t v = 0; v < nvb; v++) cptr[v + 1] += cptr[v];
        std::vector<uint32_t> crow(nnz ? nnz : 1), cidx(nnz ? nnz : 1);
        std::vector<uint64_t> cur(cptr.begin(), cptr.end() - 1);
        for (uint64_t g = 0; g < cs->num_gates; g++)
            for (uint64_t i = ptrs[k][g]; i < ptrs[k][g + 1]; i++) {
                uint32_t ci = 0;
                if (vals[k] && memcmp(vals[k] + 4 * i, &one, 32) != 0) {       // NULL vals / ONE -> slot 0
                    if (memcmp(vals[k] + 4 * i, &last, 32) == 0) ci = last_idx;
                    else {
                        std::string key((const char *)(vals[k] + 4 * i), 32);
                        auto it = dict.find(key);
                        if (it == dict.end()) { Fr v; memcpy(&v, vals[k] + 4 * i, 32); it = dict.emplace(key, (uint32_t)table.size()).first; table.push_back(v); }
                        ci = it->second; memcpy(&last, vals[k] + 4 * i, 32); last_idx = ci;
                    }
                }
                const uint64_t pos = cur[cols[k][i]]++;
                crow[pos] = (uint32_t)g; cidx[pos] = ci;
            }
        uint64_t *dp = (uint64_t *)dalloc((nvb + 1) * 8); uint32_t *dr = (uint32_t *)dalloc((nnz + 1) * 4), *di = (uint32_t *)dalloc((nnz + 1) * 4);
        d_abc[k] = (Fr *)dalloc(nv * sizeof(Fr));
        if (!dp || !dr || !di || !d_abc[k]) FK_SET_ERR(ctx, FK_ERR_OOM, "setup: device allocation failed");
        FK_HIP(ctx, hipMemcpy(dp, cptr.data(), (nvb + 1) * 8, hipMemcpyHostToDevice));
        if (nnz) { FK_HIP(ctx, hipMemcpy(dr, crow.data(), nnz * 4, hipMemcpyHostToDevice)); FK_HIP(ctx, hipMemcpy(di, cidx.data(), nnz * 4, hipMemcpyHostToDevice)); }
        ca.ptr[k] = dp; ca.row[k] = dr; ca.cidx[k] = di; ca.out[k] = d_abc[k];
        for (uint64_t v = 0; v < nvb && copies == 1; v++)
            if (cptr[v + 1] - cptr[v] > CSC_HEAVY) {
                for (uint64_t lo_ = cptr[v]; lo_ < cptr[v + 1]; lo_ += CSC_SEG) {
                    heavy_segs.push_back(HeavySeg{(uint32_t)k, 0, lo_, lo_ + CSC_SEG < cptr[v + 1] ? lo_ + CSC_SEG : cptr[v + 1]});
                    heavy_var.push_back(v);
                }
            }
    }
    Fr *d_table = (Fr *)dalloc(table.size() * sizeof(Fr));
    if (!d_table) FK_SET_ERR(ctx, FK_ERR_OOM, "setup: device allocation failed");
    FK_HIP(ctx, hipMemcpy(d_table, table.data(), table.size() * sizeof(Fr), hipMemcpyHostToDevice));

    // ---- powers of tau (two-level expansion), h scalars, Lagrange coefficients
    const uint32_t L = (log_m + 1) / 2;
    std::vector<Fr> lo((size_t)1 << L), hi((size_t)1 << (log_m - L));
    { Fr cur = Fr::one(), pw = Fr::one();
      for (size_t i = 0; i < lo.size(); i++) { lo[i] = cur; cur = Fr::mul(cur, tau); pw = Fr::mul(pw, tau); }
      cur = Fr::one();
      for (size_t j = 0; j < hi.size(); j++) { hi[j] = cur; cur = Fr::mul(cur, pw); } }
    Fr *d_lo = (Fr *)dalloc(lo.size() * sizeof(Fr)), *d_hi = (Fr *)dalloc(hi.size() * sizeof(Fr));
    Fr *d_pt = (Fr *)dalloc(m * sizeof(Fr)), *d_hs = (Fr *)dalloc(m * sizeof(Fr));
    if (!d_lo || !d_hi || !d_pt || !d_hs) FK_SET_ERR(ctx, FK_ERR_OOM, "setup: device allocation failed");
    FK_HIP(ctx, hipMemcpy(d_lo, lo.data(), lo.size() * sizeof(Fr), hipMemcpyHostToDevice));
    FK_HIP(ctx, hipMemcpy(d_hi, hi.data(), hi.size() * sizeof(Fr), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(expand_powers_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, d_lo, d_hi, L, (size_t)m, d_pt);
    const Fr delta_inv = Fr::inv(delta), gamma_inv = Fr::inv(gamma);
    const Fr coeff = Fr::mul(Fr::sub(Fr::pow_u64(tau, m), Fr::one()), delta_inv);
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, d_pt, coeff, (size_t)m, d_hs);
    FK_HIP(ctx, hipGetLastError());
    FK_TRY(ntt_exec_simple(ctx, d_pt, log_m, /*inverse=*/true, /*coset=*/false));      // d_pt := L_j(tau)

    // ---- A_k, B_k, C_k and the combined exponent
    if (copies > 1) {
        TileAsm ta;
        for (int k = 0; k < 3; k++) {
            Fr *sc = (Fr *)dalloc((size_t)copies * nvb * sizeof(Fr));
            if (!sc) FK_SET_ERR(ctx, FK_ERR_OOM, "setup: device allocation failed");
            ta.sc[k] = sc; ta.out[k] = d_abc[k]; ca.out[k] = sc;
        }
        hipLaunchKernelGGL(csc_eval_tiled_kernel, dim3((unsigned)((nvb + 255) / 256), 3, copies), dim3(256), 0, st, ca, d_table, d_pt, (uint32_t)cs->num_gates, (uint32_t)nvb);
        hipLaunchKernelGGL(tile_assemble_kernel, dim3((unsigned)((nv + 255) / 256), 3), dim3(256), 0, st, ta, d_pt, cs->num_input, cs->num_aux, copies, t_gates, num_input, nv);
        FK_HIP(ctx, hipGetLastError());
    } else
    hipLaunchKernelGGL(csc_eval_kernel, dim3((unsigned)((nv + 255) / 256), 3), dim3(256), 0, st, ca, d_table, d_pt, cs->num_gates, cs->num_input, nv);
    if (!heavy_segs.empty()) {      // heavy columns: segment partial sums on the device, folded on the host (a handful of values)
        HeavySeg *d_segs = (HeavySeg *)dalloc(heavy_segs.size() * sizeof(HeavySeg));
        Fr *d_part = (Fr *)dalloc(heavy_segs.size() * sizeof(Fr));
        if (!d_segs || !d_part) FK_SET_ERR(ctx, FK_ERR_OOM, "setup: device allocation failed");
        FK_HIP(ctx, hipMemcpyAsync(d_segs, heavy_segs.data(), heavy_segs.size() * sizeof(HeavySeg), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(csc_heavy_kernel, dim3((unsigned)heavy_segs.size()), dim3(256), 0, st, ca, d_table, d_pt, d_segs, d_part);
        FK_HIP(ctx, hipGetLastError());
        std::vector<Fr> part(heavy_segs.size());
        FK_HIP(ctx, hipMemcpyAsync(part.data(), d_part, part.size() * sizeof(Fr), hipMemcpyDeviceToHost, st));
        FK_HIP(ctx, hipStreamSynchronize(st));
        for (size_t i = 0; i < heavy_segs.size();) {
            const uint32_t mtx = heavy_segs[i].mtx; const uint64_t v = heavy_var[i];
            Fr sum = Fr::zero();
            for (; i < heavy_segs.size() && heavy_segs[i].mtx == mtx && heavy_var[i] == v; i++) sum = Fr::add(sum, part[i]);
            Fr cur;                                      // csc_eval_kernel left the light part (input row term) there
            FK_HIP(ctx, hipMemcpy(&cur, d_abc[mtx] + v, sizeof(Fr), hipMemcpyDeviceToHost));
            cur = Fr::add(cur, sum);
            FK_HIP(ctx, hipMemcpy(d_abc[mtx] + v, &cur, sizeof(Fr), hipMemcpyHostToDevice));
        }
    }
    Fr *d_e = (Fr *)dalloc(nv * sizeof(Fr));
    uint8_t *d_fa = (uint8_t *)dalloc(nv), *d_fb = (uint8_t *)dalloc(nv);
    if (!d_e || !d_fa || !d_fb) FK_SET_ERR(ctx, FK_ERR_OOM, "setup: device allocation failed");
    hipLaunchKernelGGL(combine_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, d_abc[0], d_abc[1], d_abc[2], beta, alpha, gamma_inv, delta_inv,
                       num_input, nv, d_e, d_fa, d_fb);
    FK_HIP(ctx, hipGetLastError());

    // ---- fixed-base tables (host) and the scalar multiplications
    G1Affine g1; g1.x = Fq::from_u64(1); g1.y = Fq::from_u64(2);
    G2Affine g2; { const uint32_t x0[8] = FK_G2_GEN_X0, x1[8] = FK_G2_GEN_X1, y0[8] = FK_G2_GEN_Y0, y1[8] = FK_G2_GEN_Y1;
                   for (int i = 0; i < 8; i++) { g2.x.c0.v[i] = x0[i]; g2.x.c1.v[i] = x1[i]; g2.y.c0.v[i] = y0[i]; g2.y.c1.v[i] = y1[i]; } }
    const std::vector<G1Affine> t1 = host_fb_table<Fq>(g1);
    const std::vector<G2Affine> t2 = host_fb_table<Fq2>(g2);
    G1Affine *d_t1 = (G1Affine *)dalloc(t1.size() * sizeof(G1Affine)); G2Affine *d_t2 = (G2Affine *)dalloc(t2.size() * sizeof(G2Affine));
    if (!d_t1 || !d_t2) FK_SET_ERR(ctx, FK_ERR_OOM, "setup: device allocation failed");
    FK_HIP(ctx, hipMemcpy(d_t1, t1.data(), t1.size() * sizeof(G1Affine), hipMemcpyHostToDevice));
    FK_HIP(ctx, hipMemcpy(d_t2, t2.data(), t2.size() * sizeof(G2Affine), hipMemcpyHostToDevice));

    fk_key *k = new fk_key();
    k->m = m; k->num_input = num_input; k->num_aux = num_aux; k->shard_index = shard_index; k->shard_count = shard_count;
    k->n_h = m - 1; k->n_l = num_aux;
    auto fail = [&](int code, const char *msg) { ctx->err = msg; fk_key_free(ctx, k); return code; };
    // a, b_g1, b_g2 hold only the variables whose A_v (B_v) is non-zero, in variable order: compact the SCALARS first, so
    // that the counts are known before anything is allocated and only this shard's slice of every array is ever derived
    // (W ranks deriving a key do 1/W of the fixed-base work each; the points of a variable are the same either way)
    Fr *d_sa = (Fr *)dalloc((nv + 1) * sizeof(Fr)), *d_sb = (Fr *)dalloc((nv + 1) * sizeof(Fr));
    if (!d_sa || !d_sb) return fail(FK_ERR_OOM, "setup: device allocation failed");
    uint64_t n_a = 0, n_b = 0;
    int rc = compact_elems<Fr>(ctx, d_abc[0], d_fa, nv, d_sa, &n_a);
    if (rc == FK_OK) rc = compact_elems<Fr>(ctx, d_abc[1], d_fb, nv, d_sb, &n_b);
    if (rc != FK_OK) { fk_key_free(ctx, k); return rc; }
    k->n_a = n_a; k->n_b = n_b;
    if ((rc = key_plan_slices(ctx, k, z_frac_lo, z_frac_hi)) != FK_OK) { fk_key_free(ctx, k); return rc; }
    const uint64_t c_h = k->h_hi - k->h_lo, c_l = k->l_hi - k->l_lo, c_a = k->a_hi - k->a_lo, c_b = k->b_hi - k->b_lo, c_b2 = k->b2_hi - k->b2_lo;
    G1Affine *d_ic = (G1Affine *)dalloc((size_t)num_input * sizeof(G1Affine));
    if (!d_ic) return fail(FK_ERR_OOM, "setup: device allocation failed");
    if (hipMalloc((void **)&k->d_h, (c_h + 1) * sizeof(G1Affine)) != hipSuccess || hipMalloc((void **)&k->d_l, (c_l + 1) * sizeof(G1Affine)) != hipSuccess ||
        hipMalloc((void **)&k->d_a, (c_a + 1) * sizeof(G1Affine)) != hipSuccess || hipMalloc((void **)&k->d_b1, (c_b + 1) * sizeof(G1Affine)) != hipSuccess ||
        hipMalloc((void **)&k->d_b2, (c_b2 + 1) * sizeof(G2Affine)) != hipSuccess) return fail(FK_ERR_OOM, "setup: device allocation failed");
    const unsigned fb_threads = 128;
    auto fb1 = [&](const Fr *sc, size_t n, G1Affine *o) { if (n) hipLaunchKernelGGL(HIP_KERNEL_NAME(fixed_base_kernel<Fq>), dim3((unsigned)((n + fb_threads - 1) / fb_threads)), dim3(fb_threads), 0, st, d_t1, sc, n, o); };
    fb1(d_hs + k->h_lo, c_h, k->d_h);
    fb1(d_e + num_input + k->l_lo, c_l, k->d_l);            // l = exponent points of the aux variables
    fb1(d_e, num_input, d_ic);                              // ic = those of the inputs
    fb1(d_sa + k->a_lo, c_a, k->d_a);
    fb1(d_sb + k->b_lo, c_b, k->d_b1);
    if (c_b2) hipLaunchKernelGGL(HIP_KERNEL_NAME(fixed_base_kernel<Fq2>), dim3((unsigned)((c_b2 + fb_threads - 1) / fb_threads)), dim3(fb_threads), 0, st, d_t2, d_sb + k->b2_lo, (size_t)c_b2, k->d_b2);
    if (hipGetLastError() != hipSuccess) return fail(FK_ERR_HIP, "setup: kernel launch failed");
    if (hipMemcpyAsync(ic_out, d_ic, (size_t)num_input * sizeof(G1Affine), hipMemcpyDeviceToHost, st) != hipSuccess) return fail(FK_ERR_HIP, "setup: copy failed");
    // vk
    k->alpha_g1 = host_mul<Fq>(g1, alpha); k->beta_g1 = host_mul<Fq>(g1, beta); k->delta_g1 = host_mul<Fq>(g1, delta);
    k->beta_g2 = host_mul<Fq2>(g2, beta); k->delta_g2 = host_mul<Fq2>(g2, delta);
    const G2Affine gamma_g2 = host_mul<Fq2>(g2, gamma);
    memset(vk_out, 0, 6 * 128);
    memcpy(vk_out + 0 * 128, &k->alpha_g1, 64); memcpy(vk_out + 1 * 128, &k->beta_g1, 64); memcpy(vk_out + 2 * 128, &k->beta_g2, 128);
    memcpy(vk_out + 3 * 128, &gamma_g2, 128); memcpy(vk_out + 4 * 128, &k->delta_g1, 64); memcpy(vk_out + 5 * 128, &k->delta_g2, 128);
    if (hipStreamSynchronize(st) != hipSuccess) return fail(FK_ERR_HIP, "setup: synchronize failed");
    // the scalar-side temporaries (a few GB at 2^25) go before the fixed-base levels are sized against free memory
    for (void *&p : tmp.v) { if (p) (void)hipFree(p); p = nullptr; }
    { const int rc3 = key_precompute(ctx, k); if (rc3 != FK_OK) { fk_key_free(ctx, k); return rc3; } }
    *out_key = k;
    return FK_OK;
}

int fk_setup(fk_ctx *ctx, const fk_r1cs *cs, const uint64_t tau[4], const uint64_t alpha[4], const uint64_t beta[4], const uint64_t gamma[4],
             const uint64_t delta[4], uint32_t shard_index, uint32_t shard_count, double z_frac_lo, double z_frac_hi, fk_key **out_key,
             uint8_t vk_out[6 * 128], uint8_t *ic_out) { return fk_guard(ctx, [&]() -> int {
    FK_RANGE("fk_setup");
    return setup_impl(ctx, cs, 1, tau, alpha, beta, gamma, delta, shard_index, shard_count, z_frac_lo, z_frac_hi, out_key, vk_out, ic_out);
}); }

int fk_setup_tiled(fk_ctx *ctx, const fk_r1cs *instance, uint32_t copies, const uint64_t tau[4], const uint64_t alpha[4], const uint64_t beta[4],
                   const uint64_t gamma[4], const uint64_t delta[4], uint32_t shard_index, uint32_t shard_count, double z_frac_lo, double z_frac_hi,
                   fk_key **out_key, uint8_t vk_out[6 * 128], uint8_t *ic_out) { return fk_guard(ctx, [&]() -> int {
    FK_RANGE("fk_setup_tiled");
    return setup_impl(ctx, instance, copies, tau, alpha, beta, gamma, delta, shard_index, shard_count, z_frac_lo, z_frac_hi, out_key, vk_out, ic_out);
}); }

// which: 0 = h, 1 = l, 2 = a, 3 = b_g1, 4 = b_g2.  Copies this key's slice of the array to the host.
int fk_key_download(fk_ctx *ctx, const fk_key *key, int which, void *host, size_t host_bytes) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (!key || !host) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "key download: null argument");
    FK_HIP(ctx, hipSetDevice(ctx->device));
    const void *src = nullptr; size_t bytes = 0;
    switch (which) {
        case 0: src = key->d_h; bytes = (key->h_hi - key->h_lo) * 64; break;
        case 1: src = key->d_l; bytes = (key->l_hi - key->l_lo) * 64; break;
        case 2: src = key->d_a; bytes = (key->a_hi - key->a_lo) * 64; break;
        case 3: src = key->d_b1; bytes = (key->b_hi - key->b_lo) * 64; break;
        case 4: src = key->d_b2; bytes = (key->b2_hi - key->b2_lo) * 128; break;
        default: FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "key download: which must be 0..4");
    }
    if (host_bytes < bytes) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "key download: buffer too small (%zu < %zu)", host_bytes, bytes);
    if (bytes) FK_HIP(ctx, hipMemcpy(host, src, bytes, hipMemcpyDeviceToHost));
    return FK_OK;
}); }

}  // extern "C"
