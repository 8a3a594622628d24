// A caller of kh_prove that is not Python: builds the index columns of the reference's benchmark circuit (kimchi/src/bench.rs:59-96: 2^k - 10
// generic gates `w0 - 1 = 0`, identity wiring) on the device with the C ABI alone -- the steps rust/kimchi-hip/src/prover.rs::GpuProver::new takes --,
// proves with the library's own randomness and writes the proof's sections to a file.  tests/test_gpu_native_prover.py gives it the two index
// constants a caller owns (permutation shifts, verifier-index digest), reads the proof back and has the oracle's verifier check it against the
// verifier index of the SAME circuit built independently in Python: it verifies only if these columns, commitments and the transcript agree.
// Usage: test_prove <in: log2_n | 7 shifts | digest, binary u64> <out>.  Build: g++ -std=c++17 -Iinclude tests/cpp/test_prove.cpp -lkimchi_hip
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kimchi_hip.h"

#define CK(expr) do { int rc_ = (expr); if (rc_ != KH_OK) { std::fprintf(stderr, "%s -> %d: %s\n", #expr, rc_, kh_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc != 3) { std::fprintf(stderr, "usage: test_prove in out\n"); return 2; }
    std::FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    uint64_t hdr[1 + 28 + 4];
    if (std::fread(hdr, 8, 33, f) != 33) return 2;
    std::fclose(f);
    const unsigned logn = (unsigned)hdr[0];
    const uint64_t* shifts = hdr + 1; const uint64_t* digest = hdr + 29;
    const size_t n = (size_t)1 << logn, ncol = 15 + 2 + 7 + 5, zk = 3;
    CK(kh_init(0));
    kh_srs_t* srs = nullptr;
    CK(kh_srs_create_device(KH_CURVE_VESTA, n, &srs));
    CK(kh_srs_compute_lagrange(srs, logn));
    // field arithmetic through the library's own test hooks (kh_debug_field_op: 0 mul, 1 add, 2 sub, 3 to_mont): this program has no bignum code
    enum { MUL = 0, ADD = 1, SUB = 2, TO_MONT = 3 };
    uint64_t one[4]; const uint64_t plain_one[4] = {1, 0, 0, 0};
    CK(kh_debug_field_op(KH_FIELD_FP, TO_MONT, plain_one, plain_one, one, 1));
    // ---- d1 columns on the host: coefficients (w0 - 1: c0 = 1, c4 = -1), generic selector, x (-> sid on the device), sigma filled on the device
    std::vector<uint64_t> d1(4 * ncol * n, 0);
    auto at = [&](size_t col, size_t row) { return &d1[4 * (col * n + row)]; };
    uint64_t minus_one[4];
    {
        uint64_t zero[4] = {0, 0, 0, 0};
        CK(kh_debug_field_op(KH_FIELD_FP, SUB, zero, one, minus_one, 1));
    }
    for (size_t r = 0; r + 10 < n; r++) {
        for (int k = 0; k < 4; k++) { at(0, r)[k] = one[k]; at(4, r)[k] = minus_one[k]; at(15, r)[k] = one[k]; }
    }
    for (int k = 0; k < 4; k++) at(16, 1)[k] = one[k];                     // the polynomial x: its evaluations are sid
    uint64_t *b1 = nullptr, *bc = nullptr, *b8 = nullptr;
    CK(kh_dev_alloc((void**)&b1, 32 * ncol * n)); CK(kh_dev_alloc((void**)&bc, 32 * (ncol + 2) * n)); CK(kh_dev_alloc((void**)&b8, 32 * (ncol + 2) * 8 * n));
    CK(kh_dev_upload(b1, d1.data(), 32 * ncol * n));
    CK(kh_ntt_dev(KH_FIELD_FP, b1 + 4 * 16 * n, logn, 0, 1));               // sid[j] = omega^j
    for (int i = 0; i < 7; i++) {                                           // identity wiring: sigma_i = shift_i * sid (one-token-program each)
        const uint32_t prog[6] = {KH_TOK_CELL, 0, KH_TOK_CONST, 0, KH_TOK_MUL, 0};
        const uint64_t* cols[1] = {b1 + 4 * 16 * n}; const size_t lens[1] = {n};
        CK(kh_expr_evaluations_dev(KH_FIELD_FP, prog, 3, cols, lens, 1, shifts + 4 * i, 1, n, 1, 8, 0, b1 + 4 * (17 + i) * n));
    }
    // ---- coefficient forms, then x and the permutation vanishing polynomial (x - w^(n-3))(x - w^(n-2))(x - w^(n-1)), then everything on d8
    CK(kh_dev_copy(bc, b1, 32 * ncol * n));
    CK(kh_ntt_dev(KH_FIELD_FP, bc, logn, 1, ncol));
    {
        std::vector<uint64_t> tail(4 * 2 * n, 0);
        for (int k = 0; k < 4; k++) tail[4 * 1 + k] = one[k];
        // the three roots: w^(n-3), w^(n-2), w^(n-1) = rows n-3.. of sid
        uint64_t roots[12];
        CK(kh_dev_download(roots, b1 + 4 * (16 * n + n - zk), 96));
        uint64_t ab[4], ac[4], bcx[4], abc[4], s1[4], s2[4], t[4], zero[4] = {0, 0, 0, 0};
        const uint64_t *a = roots, *b = roots + 4, *c = roots + 8;
        CK(kh_debug_field_op(KH_FIELD_FP, MUL, a, b, ab, 1)); CK(kh_debug_field_op(KH_FIELD_FP, MUL, a, c, ac, 1)); CK(kh_debug_field_op(KH_FIELD_FP, MUL, b, c, bcx, 1));
        CK(kh_debug_field_op(KH_FIELD_FP, MUL, ab, c, abc, 1));
        CK(kh_debug_field_op(KH_FIELD_FP, SUB, zero, abc, &tail[4 * n], 1));                          // -abc
        CK(kh_debug_field_op(KH_FIELD_FP, ADD, ab, ac, s1, 1)); CK(kh_debug_field_op(KH_FIELD_FP, ADD, s1, bcx, &tail[4 * (n + 1)], 1));   // ab + ac + bc
        CK(kh_debug_field_op(KH_FIELD_FP, ADD, a, b, s2, 1)); CK(kh_debug_field_op(KH_FIELD_FP, ADD, s2, c, t, 1));
        CK(kh_debug_field_op(KH_FIELD_FP, SUB, zero, t, &tail[4 * (n + 2)], 1));                       // -(a + b + c)
        for (int k = 0; k < 4; k++) tail[4 * (n + 3) + k] = one[k];
        CK(kh_dev_upload(bc + 4 * ncol * n, tail.data(), 32 * 2 * n));
    }
    CK(kh_lde_dev(KH_FIELD_FP, bc, logn, 3, b8, ncol + 2));
    kh_prover_index_t* index = nullptr;
    CK(kh_prover_index_new(srs, logn, (unsigned)zk, 0, b1, bc, b8, nullptr, 0, 0, shifts, digest, &index));
    // ---- the witness the benchmark uses: w0 = 1 on every gate row; randomness from the library
    std::vector<uint64_t> wit(4 * 15 * (n - 10), 0);
    for (size_t r = 0; r + 10 < n; r++) for (int k = 0; k < 4; k++) wit[4 * r + k] = one[k];
    kh_proof_t* proof = nullptr;
    CK(kh_prove(index, wit.data(), n - 10, nullptr, nullptr, 0, KH_PROVE_CHECK, &proof));
    std::FILE* o = std::fopen(argv[2], "wb");
    if (!o) return 2;
    for (int s = 0; s <= KH_PROOF_CHALLENGES; s++) {
        const uint64_t* limbs = nullptr; const uint8_t* flags = nullptr; size_t count = 0;
        CK(kh_proof_section(proof, s, &limbs, &flags, &count));
        const uint64_t head[2] = {(uint64_t)count, flags ? 1u : 0u};
        std::fwrite(head, 8, 2, o);
        std::fwrite(limbs, 8, (flags ? 8 : 4) * count, o);
        if (flags) std::fwrite(flags, 1, count, o);
    }
    std::fclose(o);
    double ph[6]; kh_proof_phase_seconds(proof, ph, 6);
    std::printf("PROVE_OK %.3f ms\n", 1e3 * (ph[0] + ph[1] + ph[2] + ph[3] + ph[4] + ph[5]));
    kh_proof_free(proof); kh_prover_index_free(index);
    kh_dev_free(b1); kh_dev_free(bc); kh_dev_free(b8); kh_srs_free(srs);
    return 0;
}
