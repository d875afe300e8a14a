// host_check.cpp — drives host/plonk_host.hpp (C++ on the bare C ABI, no Python in the path) and compares every result with the
// CPU oracle, loaded with dlopen (the oracle is test infrastructure: this checker lives under tests/).
//   usage: host_check <libplonk_oracle.so> <curve: 0 BN254 | 1 BLS12-381>
// Mirrors the reference's own tests: test_fft (dispatcher.rs:246-350 / dispatcher2.rs:1141-1210: 2^11 and 2^13 points, all four
// modes, S workers) and test_msm (dispatcher.rs:177-244: S contiguous shards == the monolithic MSM).
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>

#include "plonk_host.hpp"

typedef int (*orc_ntt_t)(int, uint64_t*, int, int, int, int);
typedef void (*orc_rand_fr_t)(int, uint64_t, size_t, uint64_t*);
typedef int (*orc_gen_bases_t)(int, uint64_t, size_t, size_t, uint64_t*);
typedef int (*orc_msm_t)(int, const uint64_t*, const uint8_t*, const uint64_t*, size_t, uint64_t*, int);
typedef int (*orc_jac_to_affine_t)(int, const uint64_t*, uint64_t*);
typedef int (*orc_field_op_t)(int, int, int, const uint64_t*, const uint64_t*, uint64_t*, size_t);

template <typename T> static T sym(void* h, const char* name) {
    T f = reinterpret_cast<T>(dlsym(h, name));
    if (!f) { fprintf(stderr, "missing oracle symbol %s\n", name); exit(2); }
    return f;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: host_check <oracle.so> <curve>\n"); return 2; }
    void* h = dlopen(argv[1], RTLD_NOW);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    const int curve = atoi(argv[2]);
    auto o_ntt = sym<orc_ntt_t>(h, "orc_ntt");
    auto o_rand = sym<orc_rand_fr_t>(h, "orc_rand_fr");
    auto o_bases = sym<orc_gen_bases_t>(h, "orc_gen_bases");
    auto o_msm = sym<orc_msm_t>(h, "orc_msm");
    auto o_aff = sym<orc_jac_to_affine_t>(h, "orc_jac_to_affine");
    auto o_fop = sym<orc_field_op_t>(h, "orc_field_op");
    const size_t Q = plonk::fq_limbs64(curve);
    int checks = 0;
    try {
        for (size_t S : {1, 2, 4}) {
            plonk::Dispatcher d(S, 0, curve);
            // ---- distributed NTT: odd and even log sizes (non-square r x c), the n-domain and the quotient domain
            const int log_n = 11, log_m = 13;
            const size_t n = (size_t)1 << log_n, m = (size_t)1 << log_m;
            d.init(nullptr, 0, n, m);
            for (int quot = 0; quot < 2; quot++) {
                const size_t N = quot ? m : n;
                const int log_N = quot ? log_m : log_n;
                const size_t len = N - 37;                       // shorter than the domain: the dispatcher zero-pads (:746)
                std::vector<uint64_t> coeffs(4 * N, 0);
                o_rand(curve, 1000 + S * 10 + quot, len, coeffs.data());
                for (int mode = 0; mode < 4; mode++) {
                    const bool inv = mode & 1, coset = mode & 2;
                    std::vector<uint64_t> got = d.fft(coeffs.data(), len, quot, inv, coset);
                    std::vector<uint64_t> want = coeffs;
                    if (o_ntt(curve, want.data(), log_N, inv, coset, 4)) throw plonk::Error(-9, "oracle ntt failed");
                    if (got != want) { fprintf(stderr, "FFT mismatch: S=%zu N=2^%d inv=%d coset=%d\n", S, log_N, inv, coset); return 1; }
                    checks++;
                }
            }
            // ---- sharded MSM and commit_polynomial over 2^12 + 5 bases with duplicates (P + P in buckets) and an infinity base
            const size_t nb = (1u << 12) + 5;
            std::vector<uint64_t> bases(2 * Q * nb), sc_mont(4 * nb), sc(4 * nb);
            o_bases(curve, 77, 300, nb, bases.data());
            std::vector<uint8_t> inf(nb, 0);
            for (size_t k = 0; k < 2 * Q; k++) bases[2 * Q * 3 + k] = 0;          // base 3 = infinity (dispatcher2.rs:1101), XY layout: (0, 0)
            inf[3] = 1;
            o_rand(curve, 78, nb, sc_mont.data());
            o_fop(curve, 0, 4, sc_mont.data(), nullptr, sc.data(), nb);           // into_repr
            d.init(bases.data(), nb, n, m);
            std::vector<uint64_t> jac = d.msm(sc.data(), nb);
            std::vector<uint64_t> want_jac(3 * Q), want_xy(2 * Q), got_xy(2 * Q);
            o_msm(curve, bases.data(), inf.data(), sc.data(), nb, want_jac.data(), 4);
            const int winf = o_aff(curve, want_jac.data(), want_xy.data());
            int ginf = 0;
            plonk::check(plonk_g1_to_affine(curve, jac.data(), got_xy.data(), &ginf));
            if (ginf != winf || got_xy != want_xy) { fprintf(stderr, "MSM mismatch: S=%zu\n", S); return 1; }
            std::vector<uint64_t> cxy;
            const bool cinf = d.commit_polynomial(sc_mont.data(), nb, &cxy);
            if ((int)cinf != winf || cxy != want_xy) { fprintf(stderr, "commit_polynomial mismatch: S=%zu\n", S); return 1; }
            checks += 2;
            // ---- a prover round: three polynomials of ragged length (one longer than the key: clamped like commit_polynomial, one
            // empty) through plonk_commit_many_dev on every worker's key range, against one commit_polynomial each
            {
                std::vector<uint64_t> p1(4 * (nb + 9)), p2(4 * 1000);
                o_rand(curve, 79, nb + 9, p1.data());
                o_rand(curve, 80, 1000, p2.data());
                const std::vector<const uint64_t*> polys = {sc_mont.data(), p1.data(), p2.data(), p2.data()};
                const std::vector<size_t> lens = {nb, nb + 9, 1000, 0};
                std::vector<std::vector<uint64_t>> xys;
                const std::vector<bool> infs = d.commit_round(polys, lens, &xys);
                for (size_t k = 0; k < polys.size(); k++) {
                    std::vector<uint64_t> one;
                    const bool oinf = d.commit_polynomial(polys[k], lens[k], &one);
                    if (oinf != infs[k] || (!oinf && one != xys[k])) { fprintf(stderr, "commit_round mismatch: S=%zu poly %zu\n", S, k); return 1; }
                    checks++;
                }
                if (infs[0] != (bool)winf || xys[0] != want_xy) { fprintf(stderr, "commit_round != oracle: S=%zu\n", S); return 1; }
            }
        }
    } catch (const plonk::Error& e) {
        fprintf(stderr, "plonk error %d: %s\n", e.code, e.what());
        return 1;
    }
    printf("host_check ok: %d comparisons bit-exact against the oracle (curve %d, S = 1, 2, 4)\n", checks, curve);
    return 0;
}
