// prover_check.cpp — drives host/plonk_prover.hpp: a proving key, a witness and an SRS come in through a file, the C++ prover runs the five
// rounds of `Prover::prove` (dispatcher2.rs:192-713) on the bare C ABI with its own merlin transcript, and everything it produced goes out
// through another file — no Python between this program and libplonk_hip.so.  tests/test_host_cpp.py writes the input (a satisfied circuit
// from the oracle's generator), runs this, and compares verifying key, challenges, proof and its serialization with the oracle's restatement
// of the rounds and the Python transcript.
//   usage: prover_check <in.bin> <out.bin>
// in.bin  (u64 little-endian): curve, log_n, n_bases, num_inputs, then bases[n_bases][2Q], selectors[13][n][4], sigmas[5][n][4], k[5][4],
//         wires[5][n][4], id_perm[5n][4], perm_idx[5n], pub_input[n][4], wire_blinders[5][2][4], perm_blinders[3][4]
// out.bin (u64 little-endian): points as xy[2Q] + infinity flag — 13 selector and 5 sigma commitments, 5 wire commitments, the permutation
//         commitment, the split quotient commitments (count first), opening, shifted opening; then Fr limbs: beta, gamma, alpha, zeta, v,
//         5 wire evaluations, 4 sigma evaluations, perm_next_eval; then the serialized proof (byte count, bytes padded to a multiple of 8)
#include <cstdio>
#include <cstdlib>

#include "plonk_prover.hpp"

static std::vector<uint64_t> read_all(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint64_t> v((size_t)sz / 8);
    if (fread(v.data(), 8, v.size(), f) != v.size()) { fprintf(stderr, "short read\n"); exit(2); }
    fclose(f);
    return v;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: prover_check <in.bin> <out.bin>\n"); return 2; }
    const std::vector<uint64_t> in = read_all(argv[1]);
    const int curve = (int)in[0], log_n = (int)in[1];
    const size_t n_bases = in[2], num_inputs = in[3], n = (size_t)1 << log_n, Q = plonk::fq_limbs64(curve);
    const uint64_t* p = in.data() + 4;
    auto take = [&](size_t words) { const uint64_t* q = p; p += words; return q; };
    const uint64_t* bases = take(n_bases * 2 * Q);
    const uint64_t* selectors = take(13 * n * 4);
    const uint64_t* sigmas = take(5 * n * 4);
    const uint64_t* k = take(5 * 4);
    const uint64_t* wires = take(5 * n * 4);
    const uint64_t* id_perm = take(5 * n * 4);
    const uint64_t* perm_idx = take(5 * n);
    const uint64_t* pub_input = take(n * 4);
    const uint64_t* wire_bl = take(5 * 2 * 4);
    const uint64_t* perm_bl = take(3 * 4);
    if ((size_t)(p - in.data()) != in.size()) { fprintf(stderr, "input size mismatch\n"); return 2; }
    try {
        plonk::Worker w(0, curve);
        w.init(bases, n_bases, n, 8 * n);
        plonk::Prover pv(w, log_n);
        pv.load_key(selectors, sigmas, k);
        const plonk::VerifyingKey vk = pv.verifying_key();
        plonk::Proof pr = pv.prove(wires, id_perm, perm_idx, pub_input, num_inputs, wire_bl, perm_bl, true);
        const plonk::Proof again = pv.prove(wires, id_perm, perm_idx, pub_input, num_inputs, wire_bl, perm_bl, true);      // deterministic: same bytes twice
        plonk::Codec codec(curve);
        const std::vector<uint8_t> ser = plonk::serialize_proof(codec, pr);
        if (ser != plonk::serialize_proof(codec, again)) { fprintf(stderr, "two proofs of the same witness differ\n"); return 1; }
        // an unsatisfied witness must trip the quotient-degree check (dispatcher2.rs:511-518)
        std::vector<uint64_t> bad(wires, wires + 5 * n * 4);
        bad[4 * (4 * n + 3)] ^= 1;
        bool raised = false;
        try { pv.prove(bad.data(), id_perm, perm_idx, pub_input, num_inputs, wire_bl, perm_bl, true); } catch (const plonk::WrongQuotientPolyDegree&) { raised = true; }
        if (!raised) { fprintf(stderr, "an unsatisfied witness did not raise WrongQuotientPolyDegree\n"); return 1; }

        std::vector<uint64_t> out;
        auto put_pt = [&](const plonk::Point& P) { out.insert(out.end(), P.xy.begin(), P.xy.end()); out.push_back(P.inf ? 1 : 0); };
        auto put_fr = [&](const plonk::FrEl& x) { out.insert(out.end(), x.begin(), x.end()); };
        for (const auto& P : vk.selector_comms) put_pt(P);
        for (const auto& P : vk.sigma_comms) put_pt(P);
        for (const auto& P : pr.wires_poly_comms) put_pt(P);
        put_pt(pr.prod_perm_poly_comm);
        out.push_back(pr.split_quot_poly_comms.size());
        for (const auto& P : pr.split_quot_poly_comms) put_pt(P);
        put_pt(pr.opening_proof);
        put_pt(pr.shifted_opening_proof);
        for (const char* name : {"beta", "gamma", "alpha", "zeta", "v"}) put_fr(pr.challenges.at(name));
        for (const auto& x : pr.wires_evals) put_fr(x);
        for (const auto& x : pr.wire_sigma_evals) put_fr(x);
        put_fr(pr.perm_next_eval);
        out.push_back(ser.size());
        std::vector<uint8_t> padded(ser);
        padded.resize((ser.size() + 7) / 8 * 8, 0);
        for (size_t i = 0; i < padded.size(); i += 8) {
            uint64_t wv = 0;
            for (int b = 7; b >= 0; b--) wv = (wv << 8) | padded[i + b];
            out.push_back(wv);
        }
        FILE* f = fopen(argv[2], "wb");
        if (!f || fwrite(out.data(), 8, out.size(), f) != out.size()) { fprintf(stderr, "cannot write %s\n", argv[2]); return 2; }
        fclose(f);
        printf("prover_check ok: 2^%d-gate proof, %zu serialized bytes\n", log_n, ser.size());
        (void)Q;
        return 0;
    } catch (const plonk::Error& e) {
        fprintf(stderr, "plonk error %d: %s\n", e.code, e.what());
        return 1;
    }
}
