// Sanitizer harness for the host-side gate codec (csrc/gatestream.hip: threaded decoder, threaded encoder) -- GPU sanitizers are not
// available on the pool, and this code is plain host C++ with threads, arenas and a shared dictionary: exactly what TSan / ASan are for.
// tools/sanitize/run.sh builds gatestream.hip host-only with -fsanitize=thread (then address,undefined), links this file and runs it.
// It encodes a random system (x copies) as a raw stream and through brotli, decodes both on several threads, compares the exported
// matrices with the source, and feeds the decoder truncated / corrupted streams (FK_ERR_FORMAT expected, no report from the sanitizer).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "../../include/fawkes_hip.h"

// the one symbol gatestream.hip needs from the rest of the library (fk_r1cs_load_gates -> the device loader): never reached here
struct fk_r1cs_dev;
namespace fk {
template <class P, bool I> struct Fp;
struct FrParams;
int r1cs_load_coded(fk_ctx *, unsigned, unsigned, unsigned long, const unsigned long *const *, const unsigned *const *, const unsigned *const *,
                    const Fp<FrParams, true> *, unsigned long, fk_r1cs_dev **, const unsigned char *const *) { abort(); }
std::string &tls_error();
}
extern "C" const char *fk_last_error(const fk_ctx *) { return fk::tls_error().c_str(); }       // prover.hip's, for ctx == NULL

struct Mat { std::vector<uint64_t> ptr, val; std::vector<uint32_t> col; };
#define CHECK(x) do { if (!(x)) { fprintf(stderr, "FAILED %s:%d: %s  [%s]\n", __FILE__, __LINE__, #x, fk_last_error(nullptr)); exit(1); } } while (0)

static Mat random_matrix(std::mt19937_64 &g, uint64_t gates, uint32_t nv, const std::vector<std::vector<uint64_t>> &coeffs, bool giant) {
    Mat m; m.ptr.push_back(0);
    static const uint32_t lens[] = {0, 1, 1, 1, 2, 3, 5, 9, 31, 64, 200};
    for (uint64_t i = 0; i < gates; i++) {
        uint32_t l = lens[g() % (sizeof lens / sizeof lens[0])];
        if (giant && i == gates / 2) l = 400000;            // one linear combination larger than a block
        for (uint32_t t = 0; t < l; t++) {
            m.col.push_back((uint32_t)(g() % nv));
            const auto &c = coeffs[g() % coeffs.size()];
            m.val.insert(m.val.end(), c.begin(), c.end());
        }
        m.ptr.push_back(m.col.size());
    }
    return m;
}

static void export_and_compare(fk_gates *gt, const Mat src[3], uint64_t G, uint32_t copies, uint32_t nin, uint32_t naux) {
    uint64_t info[8]; CHECK(fk_gates_info(gt, info) == FK_OK);
    CHECK(info[0] == G * copies);
    for (int k = 0; k < 3; k++) {
        const uint64_t nnz = info[1 + k];
        CHECK(nnz == src[k].col.size() * copies);
        std::vector<uint64_t> ptr(G * copies + 1), val(nnz * 4); std::vector<uint32_t> col(nnz);
        CHECK(fk_gates_export(gt, k, ptr.data(), col.data(), val.data()) == FK_OK);
        for (uint32_t c = 0; c < copies; c++) {
            for (uint64_t i = 0; i <= G; i++) CHECK(ptr[c * G + i] == c * src[k].col.size() + src[k].ptr[i]);
            for (uint64_t t = 0; t < src[k].col.size(); t++) {
                const uint32_t v = src[k].col[t];          // fk_r1cs_load_tiled's variable order: ONE, copy 0's inputs, copy 1's, ..., copy 0's aux, ...
                const uint32_t want = v == 0 ? 0 : v < nin ? 1 + c * (nin - 1) + (v - 1) : (1 + copies * (nin - 1)) + c * naux + (v - nin);
                CHECK(col[c * src[k].col.size() + t] == want);
                CHECK(memcmp(&val[(c * src[k].col.size() + t) * 4], &src[k].val[t * 4], 32) == 0);
            }
        }
    }
}

int main(int argc, char **argv) {
    const char *threads = argc > 1 ? argv[1] : "6";
    setenv("FK_HOST_THREADS", threads, 1);
    std::mt19937_64 g(2026);
    std::vector<std::vector<uint64_t>> coeffs;
    for (int i = 0; i < 300; i++) coeffs.push_back({g(), g(), g(), g() >> 4});          // < 2^252 < r: valid Montgomery residues
    const uint64_t G = 20000; const uint32_t nin = 4, naux = 18000, nv = nin + naux;
    Mat m[3] = {random_matrix(g, G, nv, coeffs, true), random_matrix(g, G, nv, coeffs, false), random_matrix(g, G, nv, coeffs, false)};
    fk_r1cs cs{}; cs.num_input = nin; cs.num_aux = naux; cs.num_gates = G;
    cs.a_ptr = m[0].ptr.data(); cs.a_col = m[0].col.data(); cs.a_val = m[0].val.data();
    cs.b_ptr = m[1].ptr.data(); cs.b_col = m[1].col.data(); cs.b_val = m[1].val.data();
    cs.c_ptr = m[2].ptr.data(); cs.c_col = m[2].col.data(); cs.c_val = m[2].val.data();
    for (uint32_t copies : {1u, 3u}) {
        const uint32_t n_in = 1 + copies * (nin - 1), n_aux = copies * naux;
        fk_blob *raw = nullptr, *br = nullptr;
        CHECK(fk_gates_encode(nullptr, &cs, copies, FK_GATES_RAW, 0, 0, &raw) == FK_OK);
        const uint8_t *rd, *bd; size_t rl, bl;
        CHECK(fk_blob_data(raw, &rd, &rl) == FK_OK);
        fprintf(stderr, "copies %u: raw stream %zu bytes\n", copies, rl);
        fk_gates *gt = nullptr;
        CHECK(fk_gates_decode(nullptr, rd, rl, FK_GATES_RAW, (uint32_t)(G * copies), n_in, n_aux, &gt) == FK_OK);
        export_and_compare(gt, m, G, copies, nin, naux);
        double prof[8]; CHECK(fk_gates_profile(gt, prof) == FK_OK);
        fprintf(stderr, "  raw decode: %.2f s, %g parse threads, %g blocks\n", prof[0], prof[5], prof[6]);
        fk_gates_free(gt); gt = nullptr;
        const int rc = fk_gates_encode(nullptr, &cs, copies, FK_GATES_BROTLI, 1, 22, &br);
        if (rc == FK_ERR_UNSUPPORTED) fprintf(stderr, "  (no libbrotlienc: brotli legs skipped)\n");
        else {
            CHECK(rc == FK_OK);
            CHECK(fk_blob_data(br, &bd, &bl) == FK_OK);
            CHECK(fk_gates_decode(nullptr, bd, bl, FK_GATES_BROTLI, (uint32_t)(G * copies), n_in, n_aux, &gt) == FK_OK);
            export_and_compare(gt, m, G, copies, nin, naux);
            CHECK(fk_gates_profile(gt, prof) == FK_OK);
            fprintf(stderr, "  brotli blob %zu bytes, decode: %.2f s, %g blocks\n", bl, prof[0], prof[6]);
            fk_gates_free(gt); gt = nullptr;
            // a blob cut short, and one with a flipped byte in the middle: an error (or, for the flip, any outcome but a crash)
            CHECK(fk_gates_decode(nullptr, bd, bl / 2, FK_GATES_BROTLI, (uint32_t)(G * copies), n_in, n_aux, &gt) == FK_ERR_FORMAT && !gt);
            std::vector<uint8_t> bad(bd, bd + bl); bad[bl / 2] ^= 0x5a;
            const int rcb = fk_gates_decode(nullptr, bad.data(), bad.size(), FK_GATES_BROTLI, (uint32_t)(G * copies), n_in, n_aux, &gt);
            if (rcb == FK_OK) fk_gates_free(gt);
            gt = nullptr;
            fk_blob_free(br);
        }
        // malformed raw streams: truncated, trailing byte, bad tag / index out of range far apart (the earliest error must be reported)
        CHECK(fk_gates_decode(nullptr, rd, rl - 5, FK_GATES_RAW, (uint32_t)(G * copies), n_in, n_aux, &gt) == FK_ERR_FORMAT && !gt);
        std::vector<uint8_t> bad(rd, rd + rl); bad.push_back(0);
        CHECK(fk_gates_decode(nullptr, bad.data(), bad.size(), FK_GATES_RAW, (uint32_t)(G * copies), n_in, n_aux, &gt) == FK_ERR_FORMAT && !gt);
        bad.pop_back();
        for (size_t pos : {rl / 5, rl / 2, rl - rl / 7}) for (size_t d = 0; d < 64; d++) bad[pos + d] ^= 0xff;
        CHECK(fk_gates_decode(nullptr, bad.data(), bad.size(), FK_GATES_RAW, (uint32_t)(G * copies), n_in, n_aux, &gt) == FK_ERR_FORMAT && !gt);
        fprintf(stderr, "  malformed: %s\n", fk_last_error(nullptr));
        // wrong variable counts: an index out of range
        CHECK(fk_gates_decode(nullptr, rd, rl, FK_GATES_RAW, (uint32_t)(G * copies), n_in, n_aux / 2, &gt) == FK_ERR_FORMAT && !gt);
        fk_blob_free(raw);
    }
    fprintf(stderr, "gatestream harness: ok (%s threads)\n", threads);
    return 0;
}
