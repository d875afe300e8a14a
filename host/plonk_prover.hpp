// plonk_prover.hpp — the reference prover's five rounds in C++, on nothing but the C ABI of include/plonk_hip.h.
//
// What it mirrors (reference /root/reference/src): `Prover::prove` (dispatcher2.rs:192-713) with its Fiat-Shamir transcript
// `FakeStandardTranscript` (dispatcher2.rs:44-154: a merlin::Transcript — STROBE-128 over Keccak-f[1600] — fed with ark-serialize's
// compressed encodings) and the proof it assembles (:699-710).  The reference's host side is compiled code (Rust; no toolchain in this
// image — ffi/plonk_hip.rs is its binding as source); this header is the same orchestration as compiled code: a C++ program links
// libplonk_hip.so, loads a proving key and a witness, and gets a proof without Python anywhere (tests/host_cpp/prover_check.cpp).
// Same structure as distributed_plonk_amd/prover.py (the Python mirror the bench uses): every vector stays in HBM, only commitments,
// evaluations and challenges cross the host boundary; scalar (challenge) arithmetic is host code here too (dispatcher2.rs:558-646
// works on single Fr values).
//
//   plonk::Modulus<N>        N x u64 Montgomery arithmetic on the host (Fr: N = 4; Fq: 4 / 6 for the point encodings)
//   plonk::Merlin            merlin::Transcript (new / append_message / challenge_bytes)
//   plonk::PlonkTranscript   dispatcher2.rs:44-154, method for method
//   plonk::Prover            load_key / verifying_key / prove  (dispatcher2.rs:238-241, 296-712)
// Points are affine x || y Montgomery limbs + an infinity flag (what plonk_g1_to_affine returns); Fr are 4 x u64 Montgomery limbs.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "plonk_host.hpp"

namespace plonk {

typedef unsigned __int128 u128;

// ---------------------------------------------------------------------------------------------- host modular arithmetic
template <int N> struct Modulus {
    std::array<uint64_t, N> p{}, r2{}, one{};
    uint64_t inv = 0;                                   // -p^-1 mod 2^64
    int bits = 0;
    typedef std::array<uint64_t, N> El;

    explicit Modulus(const std::array<uint64_t, N>& modulus) : p(modulus) {
        uint64_t x = 1;
        for (int i = 0; i < 6; i++) x *= 2 - p[0] * x;  // Newton: p^-1 mod 2^64
        inv = (uint64_t)0 - x;
        for (int i = N - 1; i >= 0 && bits == 0; i--)
            for (int b = 63; b >= 0; b--)
                if ((p[i] >> b) & 1) { bits = 64 * i + b + 1; break; }
        El t{};                                         // 2^(64N) mod p by 64N doublings of 1, then squared-by-doubling once more for R^2
        t[0] = 1;
        for (int i = 0; i < 64 * N; i++) t = dbl(t);
        one = t;
        for (int i = 0; i < 64 * N; i++) t = dbl(t);
        r2 = t;
    }
    static bool geq(const El& a, const El& b) {
        for (int i = N - 1; i >= 0; i--) { if (a[i] != b[i]) return a[i] > b[i]; }
        return true;
    }
    static El sub_raw(const El& a, const El& b) {
        El r{};
        uint64_t br = 0;
        for (int i = 0; i < N; i++) {
            const u128 t = (u128)a[i] - b[i] - br;
            r[i] = (uint64_t)t;
            br = (uint64_t)(t >> 64) & 1;
        }
        return r;
    }
    El add(const El& a, const El& b) const {
        El r{};
        uint64_t c = 0;
        for (int i = 0; i < N; i++) { const u128 t = (u128)a[i] + b[i] + c; r[i] = (uint64_t)t; c = (uint64_t)(t >> 64); }
        if (c || geq(r, p)) r = sub_raw(r, p);
        return r;
    }
    El dbl(const El& a) const { return add(a, a); }
    El sub(const El& a, const El& b) const {
        if (geq(a, b)) return sub_raw(a, b);
        El t = sub_raw(b, a);
        return sub_raw(p, t);
    }
    El neg(const El& a) const { El z{}; return sub(z, a); }
    // Montgomery product a * b / 2^(64N) mod p (CIOS)
    El mul(const El& a, const El& b) const {
        uint64_t t[N + 2] = {0};
        for (int i = 0; i < N; i++) {
            u128 c = 0;
            for (int j = 0; j < N; j++) { c += (u128)a[j] * b[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
            c += t[N]; t[N] = (uint64_t)c; t[N + 1] = (uint64_t)(c >> 64);
            const uint64_t m = t[0] * inv;
            c = (u128)m * p[0] + t[0];
            c >>= 64;
            for (int j = 1; j < N; j++) { c += (u128)m * p[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
            c += t[N]; t[N - 1] = (uint64_t)c; t[N] = t[N + 1] + (uint64_t)(c >> 64);
        }
        El r{};
        for (int i = 0; i < N; i++) r[i] = t[i];
        if (t[N] || geq(r, p)) r = sub_raw(r, p);
        return r;
    }
    El sqr(const El& a) const { return mul(a, a); }
    El to_mont(const El& canonical) const { return mul(canonical, r2); }
    El from_mont(const El& a) const { El o{}; o[0] = 1; return mul(a, o); }
    El from_u64(uint64_t v) const { El c{}; c[0] = v; return to_mont(c); }
    // a^e for a plain (non-Montgomery) exponent given as limbs
    El pow(const El& a, const El& e) const {
        El r = one;
        for (int i = 64 * N - 1; i >= 0; i--) {
            r = sqr(r);
            if ((e[i / 64] >> (i % 64)) & 1) r = mul(r, a);
        }
        return r;
    }
    El pow_u64(const El& a, uint64_t e) const { El x{}; x[0] = e; return pow(a, x); }
    El inverse(const El& a) const {                     // Fermat; the reference's Fp division unwraps the inverse (a = 0 is the caller's bug)
        El e = p;
        El two{}; two[0] = 2;
        e = sub_raw(e, two);
        return pow(a, e);
    }
    bool is_zero(const El& a) const { for (int i = 0; i < N; i++) if (a[i]) return false; return true; }
};

struct FrParams {
    std::array<uint64_t, 4> modulus;
    uint64_t generator;
    int two_adicity;
};
inline FrParams fr_params_of(int curve) {
    if (curve == PLONK_BN254)
        return {{0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull}, 5, 28};
    return {{0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull}, 7, 32};
}
inline std::array<uint64_t, 4> fq_modulus_bn254() {
    return {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
}
inline std::array<uint64_t, 6> fq_modulus_bls12_381() {
    return {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull, 0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
}

typedef std::array<uint64_t, 4> FrEl;

// Fr with the domain helpers the prover needs (ark-ff FftField::get_root_of_unity, multiplicative_generator)
struct FrField : Modulus<4> {
    uint64_t generator;
    int two_adicity;
    explicit FrField(int curve) : Modulus<4>(fr_params_of(curve).modulus), generator(fr_params_of(curve).generator), two_adicity(fr_params_of(curve).two_adicity) {}
    FrEl gen() const { return from_u64(generator); }
    // the 2^s-th root g^((p-1)/2^s) squared down to order n
    FrEl root_of_unity(uint64_t n) const {
        int log = 0;
        while (((uint64_t)1 << log) < n) log++;
        if (((uint64_t)1 << log) != n || log > two_adicity) throw Error(PLONK_ERR_DOMAIN, "DomainCreationError");
        FrEl e = p;                                     // (p - 1) >> two_adicity
        e[0] -= 1;
        for (int s = 0; s < two_adicity; s++) {
            for (int i = 0; i < 4; i++) e[i] = (e[i] >> 1) | (i < 3 ? e[i + 1] << 63 : 0);
        }
        FrEl w = pow(gen(), e);
        for (int i = 0; i < two_adicity - log; i++) w = sqr(w);
        return w;
    }
};

// ---------------------------------------------------------------------------------------------- Keccak-f[1600], STROBE-128, Merlin
inline uint64_t rol64(uint64_t v, int r) { r &= 63; return r ? (v << r) | (v >> (64 - r)) : v; }
inline void keccak_f1600(uint8_t state[200]) {
    static const uint64_t RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull, 0x000000000000808Bull,
                                    0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008Aull, 0x0000000000000088ull,
                                    0x0000000080008009ull, 0x000000008000000Aull, 0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull,
                                    0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
                                    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    static const int ROT[5][5] = {{0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61}, {28, 55, 25, 21, 56}, {27, 20, 39, 8, 14}};   // [x][y]
    uint64_t a[5][5];
    for (int x = 0; x < 5; x++)
        for (int y = 0; y < 5; y++) {
            uint64_t v = 0;
            for (int b = 7; b >= 0; b--) v = (v << 8) | state[8 * (x + 5 * y) + b];
            a[x][y] = v;
        }
    for (int rnd = 0; rnd < 24; rnd++) {
        uint64_t c[5], d[5], b[5][5];
        for (int x = 0; x < 5; x++) c[x] = a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4];
        for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) a[x][y] ^= d[x];
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) b[y][(2 * x + 3 * y) % 5] = rol64(a[x][y], ROT[x][y]);
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) a[x][y] = b[x][y] ^ (~b[(x + 1) % 5][y] & b[(x + 2) % 5][y]);
        a[0][0] ^= RC[rnd];
    }
    for (int x = 0; x < 5; x++)
        for (int y = 0; y < 5; y++)
            for (int b = 0; b < 8; b++) state[8 * (x + 5 * y) + b] = (uint8_t)(a[x][y] >> (8 * b));
}

class Strobe128 {                                       // the subset Merlin uses
  public:
    explicit Strobe128(const std::string& protocol_label) {
        std::memset(st_, 0, sizeof st_);
        const uint8_t head[6] = {1, R + 2, 1, 0, 1, 96};
        std::memcpy(st_, head, 6);
        std::memcpy(st_ + 6, "STROBEv1.0.2", 12);
        keccak_f1600(st_);
        meta_ad(bytes_of(protocol_label), false);
    }
    static std::vector<uint8_t> bytes_of(const std::string& s) { return std::vector<uint8_t>(s.begin(), s.end()); }
    void meta_ad(const std::vector<uint8_t>& data, bool more) { begin_op(FLAG_M | FLAG_A, more); absorb(data); }
    void ad(const std::vector<uint8_t>& data, bool more) { begin_op(FLAG_A, more); absorb(data); }
    std::vector<uint8_t> prf(size_t n, bool more) { begin_op(FLAG_I | FLAG_A | FLAG_C, more); return squeeze(n); }

  private:
    static constexpr uint8_t R = 166;
    static constexpr uint8_t FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_T = 8, FLAG_M = 16, FLAG_K = 32;
    uint8_t st_[200];
    uint8_t pos_ = 0, pos_begin_ = 0, cur_flags_ = 0;
    void run_f() {
        st_[pos_] ^= pos_begin_;
        st_[pos_ + 1] ^= 0x04;
        st_[R + 1] ^= 0x80;
        keccak_f1600(st_);
        pos_ = pos_begin_ = 0;
    }
    void absorb(const std::vector<uint8_t>& data) {
        for (uint8_t byte : data) {
            st_[pos_] ^= byte;
            if (++pos_ == R) run_f();
        }
    }
    std::vector<uint8_t> squeeze(size_t n) {
        std::vector<uint8_t> out(n);
        for (size_t i = 0; i < n; i++) {
            out[i] = st_[pos_];
            st_[pos_] = 0;
            if (++pos_ == R) run_f();
        }
        return out;
    }
    void begin_op(uint8_t flags, bool more) {
        if (more) {
            if (cur_flags_ != flags) throw Error(PLONK_ERR_STATE, "strobe: continued operation with different flags");
            return;
        }
        if (flags & FLAG_T) throw Error(PLONK_ERR_ARG, "strobe: transport operations are not used by merlin");
        const uint8_t old_begin = pos_begin_;
        pos_begin_ = pos_ + 1;
        cur_flags_ = flags;
        absorb({old_begin, flags});
        if ((flags & (FLAG_C | FLAG_K)) && pos_ != 0) run_f();
    }
};

class Merlin {                                          // merlin::Transcript 3.0.0
  public:
    explicit Merlin(const std::string& label) : strobe_("Merlin v1.0") { append_message("dom-sep", Strobe128::bytes_of(label)); }
    void append_message(const std::string& label, const std::vector<uint8_t>& message) {
        strobe_.meta_ad(Strobe128::bytes_of(label), false);
        strobe_.meta_ad(le32((uint32_t)message.size()), true);
        strobe_.ad(message, false);
    }
    std::vector<uint8_t> challenge_bytes(const std::string& label, size_t n) {
        strobe_.meta_ad(Strobe128::bytes_of(label), false);
        strobe_.meta_ad(le32((uint32_t)n), true);
        return strobe_.prf(n, false);
    }

  private:
    Strobe128 strobe_;
    static std::vector<uint8_t> le32(uint32_t v) { return {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24)}; }
};

// ---------------------------------------------------------------------------------------------- ark-serialize 0.3 (compressed) + the transcript
struct Point {                                          // what plonk_g1_to_affine returns
    std::vector<uint64_t> xy;                           // x || y, Montgomery, 2 * Q limbs
    bool inf = false;
};

inline std::vector<uint8_t> le64(uint64_t v) {
    std::vector<uint8_t> o(8);
    for (int i = 0; i < 8; i++) o[i] = (uint8_t)(v >> (8 * i));
    return o;
}
template <int N> inline std::vector<uint8_t> le_bytes(const std::array<uint64_t, N>& v) {
    std::vector<uint8_t> o(8 * N);
    for (int i = 0; i < 8 * N; i++) o[i] = (uint8_t)(v[i / 8] >> (8 * (i % 8)));
    return o;
}

class Codec {                                           // `jf_utils::to_bytes!` = CanonicalSerialize::serialize (compressed)
  public:
    explicit Codec(int curve) : curve_(curve), fr_(curve), q4_(fq_modulus_bn254()), q6_(fq_modulus_bls12_381()) {}
    const FrField& fr() const { return fr_; }
    // Fr: the canonical (non-Montgomery) integer, 32 bytes little-endian
    std::vector<uint8_t> fr_bytes(const FrEl& mont) const { return le_bytes<4>(fr_.from_mont(mont)); }
    // G1Affine: x little-endian with SWFlags in the two top bits of the last byte (bit 7: y > -y, bit 6: infinity)
    std::vector<uint8_t> g1_bytes(const Point& P) const { return curve_ == PLONK_BN254 ? g1_bytes_t<4>(q4_, P) : g1_bytes_t<6>(q6_, P); }

  private:
    int curve_;
    FrField fr_;
    Modulus<4> q4_;
    Modulus<6> q6_;
    template <int N> static std::vector<uint8_t> g1_bytes_t(const Modulus<N>& q, const Point& P) {
        std::vector<uint8_t> out(8 * N, 0);
        if (P.inf) { out.back() |= 1 << 6; return out; }
        std::array<uint64_t, N> x{}, y{};
        for (int i = 0; i < N; i++) { x[i] = P.xy[i]; y[i] = P.xy[N + i]; }
        const auto xc = q.from_mont(x), yc = q.from_mont(y), ny = q.neg(yc);
        out = le_bytes<N>(xc);
        if (!Modulus<N>::geq(ny, yc)) out.back() |= 1 << 7;      // y > -y
        return out;
    }
};

class PlonkTranscript {                                 // dispatcher2.rs:44-154 (`FakeStandardTranscript`)
  public:
    explicit PlonkTranscript(int curve, const std::string& label = "PlonkProof") : codec_(curve), t_(label) {}   // :238
    const Codec& codec() const { return codec_; }
    void append_vk_and_pub_input(uint64_t domain_size, const std::vector<FrEl>& k, const std::vector<Point>& selector_comms, const std::vector<Point>& sigma_comms,
                                 const std::vector<FrEl>& pub_input) {                                         // :56-91
        t_.append_message("field size in bits", le64((uint64_t)codec_.fr().bits));
        t_.append_message("domain size", le64(domain_size));
        t_.append_message("input size", le64((uint64_t)pub_input.size()));
        for (const FrEl& ki : k) t_.append_message("wire subsets separators", codec_.fr_bytes(ki));
        for (const Point& c : selector_comms) t_.append_message("selector commitments", codec_.g1_bytes(c));
        for (const Point& c : sigma_comms) t_.append_message("sigma commitments", codec_.g1_bytes(c));
        for (const FrEl& x : pub_input) t_.append_message("public input", codec_.fr_bytes(x));
    }
    void append_commitments(const std::string& label, const std::vector<Point>& comms) { for (const Point& c : comms) t_.append_message(label, codec_.g1_bytes(c)); }   // :94-107
    void append_commitment(const std::string& label, const Point& c) { t_.append_message(label, codec_.g1_bytes(c)); }                                                // :110-121
    void append_proof_evaluations(const std::vector<FrEl>& wires_evals, const std::vector<FrEl>& wire_sigma_evals, const FrEl& perm_next_eval) {                       // :124-138
        for (const FrEl& e : wires_evals) t_.append_message("wire_evals", codec_.fr_bytes(e));
        for (const FrEl& e : wire_sigma_evals) t_.append_message("wire_sigma_evals", codec_.fr_bytes(e));
        t_.append_message("perm_next_eval", codec_.fr_bytes(perm_next_eval));
    }
    // :142-153: 64 transcript bytes reduced mod r (from_le_bytes_mod_order), appended back as 32 canonical bytes -> Montgomery limbs
    FrEl get_and_append_challenge(const std::string& label) {
        const std::vector<uint8_t> buf = t_.challenge_bytes(label, 64);
        const FrField& f = codec_.fr();
        // value = lo + hi * 2^256 with lo, hi the two 32-byte halves; every piece goes through Montgomery form: x -> x * R2 / R = x * R mod p
        FrEl lo{}, hi{};
        for (int i = 0; i < 32; i++) { lo[i / 8] |= (uint64_t)buf[i] << (8 * (i % 8)); hi[i / 8] |= (uint64_t)buf[32 + i] << (8 * (i % 8)); }
        // to_mont reduces any 256-bit input (CIOS accepts operands below 2^256): mont(lo) + mont(hi) * mont(2^256) = mont(lo + hi * 2^256)
        const FrEl two256 = f.to_mont(f.one);            // one = 2^256 mod p as a plain residue -> its Montgomery form
        const FrEl c = f.add(f.to_mont(lo), f.mul(f.to_mont(hi), two256));
        t_.append_message(label, codec_.fr_bytes(c));
        return c;
    }

  private:
    Codec codec_;
    Merlin t_;
};

// ---------------------------------------------------------------------------------------------- the proof and the prover
struct Proof {                                          // the fields of `Proof` (dispatcher2.rs:699-710)
    std::vector<Point> wires_poly_comms, split_quot_poly_comms;
    Point prod_perm_poly_comm, opening_proof, shifted_opening_proof;
    std::vector<FrEl> wires_evals, wire_sigma_evals;
    FrEl perm_next_eval{};
    std::map<std::string, FrEl> challenges;             // beta, gamma, alpha, zeta, v as drawn (Montgomery limbs)
};

struct VerifyingKey {                                   // what the transcript absorbs of jf-plonk's VerifyingKey (dispatcher2.rs:56-91)
    uint64_t domain_size = 0;
    std::vector<FrEl> k;
    std::vector<Point> selector_comms, sigma_comms;
};

struct WrongQuotientPolyDegree : Error {                // SnarkError::WrongQuotientPolyDegree, dispatcher2.rs:513-517
    int64_t got, expected;
    WrongQuotientPolyDegree(int64_t g, int64_t e) : Error(PLONK_ERR_ARG, "WrongQuotientPolyDegree(" + std::to_string(g) + ", " + std::to_string(e) + ")"), got(g), expected(e) {}
};

// `serialize` of the Proof struct in the field order of dispatcher2.rs:699-710 (Vec = u64 length + elements).  UNVERIFIED LAYOUT, like
// transcript.py's serialize_proof: the struct lives in an un-vendored crate; every ELEMENT's encoding is what the transcript absorbs.
inline std::vector<uint8_t> serialize_proof(const Codec& c, const Proof& pr) {
    std::vector<uint8_t> out;
    auto put = [&](const std::vector<uint8_t>& b) { out.insert(out.end(), b.begin(), b.end()); };
    auto vec_pts = [&](const std::vector<Point>& v) { put(le64(v.size())); for (const Point& p : v) put(c.g1_bytes(p)); };
    auto vec_fr = [&](const std::vector<FrEl>& v) { put(le64(v.size())); for (const FrEl& x : v) put(c.fr_bytes(x)); };
    vec_pts(pr.wires_poly_comms);
    put(c.g1_bytes(pr.prod_perm_poly_comm));
    vec_pts(pr.split_quot_poly_comms);
    put(c.g1_bytes(pr.opening_proof));
    put(c.g1_bytes(pr.shifted_opening_proof));
    vec_fr(pr.wires_evals);
    vec_fr(pr.wire_sigma_evals);
    put(c.fr_bytes(pr.perm_next_eval));
    return out;
}

// One GPU.  The worker must have been `init`-ed with the commit key padded as dispatcher2.rs:207-208 does and the domains n, 8n.
class Prover {
  public:
    static constexpr int NUM_WIRE_TYPES = 5, NUM_SELECTORS = 13;
    Prover(Worker& w, int log_n) : w_(w), ctx_(w.ctx()), curve_(w.curve()), f_(w.curve()), log_n_(log_n), n_((size_t)1 << log_n), m_((size_t)8 << log_n) {}
    ~Prover() { for (void* p : bufs_) plonk_dev_free(ctx_, p); }
    Prover(const Prover&) = delete;
    Prover& operator=(const Prover&) = delete;

    // ProvingKey polynomials in coefficient form: selectors 13 x n, sigmas 5 x n (host, Montgomery); k = vk.k (5)
    void load_key(const uint64_t* selectors, const uint64_t* sigmas, const uint64_t* k) {
        sel_ = alloc(NUM_SELECTORS * n_);
        sig_ = alloc(NUM_WIRE_TYPES * n_);
        check(plonk_memcpy_h2d(ctx_, sel_, selectors, NUM_SELECTORS * n_ * 32));
        check(plonk_memcpy_h2d(ctx_, sig_, sigmas, NUM_WIRE_TYPES * n_ * 32));
        k_.assign(NUM_WIRE_TYPES, FrEl{});
        for (int i = 0; i < NUM_WIRE_TYPES; i++) std::memcpy(k_[i].data(), k + 4 * i, 32);
        have_vk_ = false;
    }
    // commitments of the 13 selector and 5 sigma polynomials (once per key)
    const VerifyingKey& verifying_key() {
        if (!have_vk_) {
            vk_ = VerifyingKey();
            vk_.domain_size = n_;
            vk_.k = k_;
            for (int i = 0; i < NUM_SELECTORS; i++) vk_.selector_comms.push_back(commit(at(sel_, i * n_), n_));
            for (int i = 0; i < NUM_WIRE_TYPES; i++) vk_.sigma_comms.push_back(commit(at(sig_, i * n_), n_));
            have_vk_ = true;
        }
        return vk_;
    }

    // wires 5 x n: witness[wire_variables[i][j]]; id_perm 5n: extended_id_permutation; perm_idx 5n u64: perm_i*n+perm_j; pub_input n
    // evaluations (zero-padded), of which the first num_inputs are the public inputs the transcript absorbs; blinders: wires 5 x 2, perm 3.
    Proof prove(const uint64_t* wires, const uint64_t* id_perm, const uint64_t* perm_idx, const uint64_t* pub_input, size_t num_inputs,
                const uint64_t* wire_blinders, const uint64_t* perm_blinders, bool check_degree = true) {
        if (!sel_) throw Error(PLONK_ERR_STATE, "load_key first");
        const size_t n = n_, m = m_;
        const FrField& f = f_;
        Proof proof;
        PlonkTranscript t(curve_);
        {                                               // :238-241
            const VerifyingKey& vk = verifying_key();
            std::vector<FrEl> pi(num_inputs);
            for (size_t i = 0; i < num_inputs; i++) std::memcpy(pi[i].data(), pub_input + 4 * i, 32);
            t.append_vk_and_pub_input(vk.domain_size, vk.k, vk.selector_comms, vk.sigma_comms, pi);
        }
        Scope scope(*this);                             // frees this proof's device buffers on every exit
        void* d_wev = scope.alloc(5 * n);
        void* d_id = scope.alloc(5 * n);
        void* d_idx = scope.alloc((5 * n + 3) / 4);
        void* d_pi = scope.alloc(n);
        check(plonk_memcpy_h2d(ctx_, d_wev, wires, 5 * n * 32));
        check(plonk_memcpy_h2d(ctx_, d_id, id_perm, 5 * n * 32));
        check(plonk_memcpy_h2d(ctx_, d_idx, perm_idx, 5 * n * 8));
        check(plonk_memcpy_h2d(ctx_, d_pi, pub_input, n * 32));
        const void* wev[5];
        for (int i = 0; i < 5; i++) wev[i] = at(d_wev, i * n);

        // ---- Round 1 (:296-322): wire polynomials and their commitments
        const size_t WP = n + 2;
        void* d_wp = scope.alloc(5 * WP);
        void* d_tmp = scope.alloc(n);
        for (int i = 0; i < 5; i++) {
            check(plonk_memset_dev(ctx_, at(d_wp, i * WP + n), 0, (WP - n) * 32));          // the interpolation writes coefficients 0 .. n-1
            interpolate(d_tmp, wev[i], at(d_wp, i * WP));
            check(plonk_blind_dev(ctx_, at(d_wp, i * WP), n, wire_blinders + 8 * i, 2));
        }
        {
            std::vector<std::pair<const void*, size_t>> items;
            for (int i = 0; i < 5; i++) items.push_back({at(d_wp, i * WP), WP});
            proof.wires_poly_comms = commit_many(items);
        }
        // ---- Round 2 (:325-357): permutation product polynomial
        t.append_commitments("witness_poly_comms", proof.wires_poly_comms);
        const FrEl beta = proof.challenges["beta"] = t.get_and_append_challenge("beta");
        const FrEl gamma = proof.challenges["gamma"] = t.get_and_append_challenge("gamma");
        void* d_prod = scope.alloc(n);
        check(plonk_perm_product_dev(ctx_, wev, d_id, d_idx, beta.data(), gamma.data(), n, d_prod));
        const size_t PP = n + 3;
        void* d_pp = scope.alloc(PP);
        check(plonk_memset_dev(ctx_, at(d_pp, n), 0, (PP - n) * 32));
        interpolate(d_tmp, d_prod, d_pp);
        check(plonk_blind_dev(ctx_, d_pp, n, perm_blinders, 3));
        proof.prod_perm_poly_comm = commit(d_pp, PP);
        // ---- Round 3 (:360-533): quotient polynomial
        t.append_commitment("perm_poly_comms", proof.prod_perm_poly_comm);
        const FrEl alpha = proof.challenges["alpha"] = t.get_and_append_challenge("alpha");
        void* d_pi_poly = scope.alloc(n);
        interpolate(d_tmp, d_pi, d_pi_poly);                                     // :426
        void* d_kc = scope.alloc(18 * m);
        void* d_c = scope.alloc(7 * m);
        const FrEl g = f.gen();
        for (int j = 0; j < 18; j++)                                             // :382-405: selectors, sigmas
            check(plonk_coset_eval_dev(ctx_, j < 13 ? at(sel_, j * n) : at(sig_, (j - 13) * n), n, m, g.data(), at(d_kc, j * m)));
        for (int j = 0; j < 5; j++) check(plonk_coset_eval_dev(ctx_, at(d_wp, j * WP), WP, m, g.data(), at(d_c, j * m)));     // :406-424
        check(plonk_coset_eval_dev(ctx_, d_pp, PP, m, g.data(), at(d_c, 5 * m)));
        check(plonk_coset_eval_dev(ctx_, d_pi_poly, n, m, g.data(), at(d_c, 6 * m)));
        plonk_quotient_inputs qi;
        for (int j = 0; j < 13; j++) qi.selectors[j] = at(d_kc, j * m);
        for (int j = 0; j < 5; j++) { qi.sigmas[j] = at(d_kc, (13 + j) * m); qi.wires[j] = at(d_c, j * m); }
        qi.perm = at(d_c, 5 * m);
        qi.pub_input = at(d_c, 6 * m);
        void* d_qev = scope.alloc(m);
        std::vector<uint64_t> kflat(4 * NUM_WIRE_TYPES);
        for (int i = 0; i < NUM_WIRE_TYPES; i++) std::memcpy(&kflat[4 * i], k_[i].data(), 32);
        check(plonk_quotient_evals_dev(ctx_, &qi, alpha.data(), beta.data(), gamma.data(), kflat.data(), d_qev));            // :435-504
        void* d_quot = scope.alloc(m);
        check(plonk_ntt_dev(ctx_, d_qev, d_quot, m, 1, 1));                      // :507
        const int64_t expected = (int64_t)NUM_WIRE_TYPES * (int64_t)(n + 1) + 2;
        if (check_degree) {                                                      // :511-518
            int64_t deg = 0;
            check(plonk_poly_degree_dev(ctx_, d_quot, m, &deg));
            if (deg != expected) throw WrongQuotientPolyDegree(deg, expected);
        }
        std::vector<std::pair<const void*, size_t>> split;                       // coeffs.chunks(n + 2)  (:519-523)
        for (size_t off = 0; off < (size_t)expected + 1; off += n + 2) split.push_back({at(d_quot, off), std::min(n + 2, (size_t)expected + 1 - off)});
        proof.split_quot_poly_comms = commit_many(split);
        // ---- Round 4 (:536-555): evaluations at zeta
        t.append_commitments("quot_poly_comms", proof.split_quot_poly_comms);
        const FrEl zeta = proof.challenges["zeta"] = t.get_and_append_challenge("zeta");
        const FrEl zeta_w = f.mul(zeta, f.root_of_unity(n));
        for (int i = 0; i < 5; i++) proof.wires_evals.push_back(eval(at(d_wp, i * WP), WP, zeta));
        for (int i = 0; i < 4; i++) proof.wire_sigma_evals.push_back(eval(at(sig_, i * n), n, zeta));
        proof.perm_next_eval = eval(d_pp, PP, zeta_w);
        // ---- Round 5 (:558-690): linearisation polynomial, batched opening, shifted opening
        t.append_proof_evaluations(proof.wires_evals, proof.wire_sigma_evals, proof.perm_next_eval);
        const FrEl v = proof.challenges["v"] = t.get_and_append_challenge("v");
        const FrEl &a = proof.wires_evals[0], &b = proof.wires_evals[1], &c = proof.wires_evals[2], &d = proof.wires_evals[3], &e = proof.wires_evals[4];
        const FrEl vanish = f.sub(f.pow_u64(zeta, n), f.one);
        const FrEl ab = f.mul(a, b), cd = f.mul(c, d);
        auto pow5 = [&](const FrEl& x) { const FrEl x2 = f.sqr(x); return f.mul(f.sqr(x2), x); };
        std::vector<std::pair<const void*, size_t>> polys;
        std::vector<FrEl> coeffs = {a, b, c, d, ab, cd, pow5(a), pow5(b), pow5(c), pow5(d), f.neg(e), f.one, f.mul(f.mul(ab, cd), e)};     // :566-600 (selector order :443-456)
        for (int j = 0; j < 13; j++) polys.push_back({at(sel_, j * n), n});
        const FrEl l1 = f.mul(vanish, f.inverse(f.mul(f.from_u64(n), f.sub(zeta, f.one))));        // L_1(zeta) = (zeta^n - 1) / (n (zeta - 1))
        FrEl acc = alpha;
        for (int i = 0; i < 5; i++) acc = f.mul(acc, f.add(f.add(proof.wires_evals[i], f.mul(f.mul(beta, k_[i]), zeta)), gamma));
        polys.push_back({d_pp, PP});
        coeffs.push_back(f.add(acc, f.mul(f.sqr(alpha), l1)));
        acc = f.mul(f.mul(alpha, beta), proof.perm_next_eval);
        for (int i = 0; i < 4; i++) acc = f.mul(acc, f.add(f.add(proof.wires_evals[i], f.mul(beta, proof.wire_sigma_evals[i])), gamma));
        polys.push_back({at(sig_, 4 * n), n});
        coeffs.push_back(f.neg(acc));
        const FrEl z_n2 = f.mul(f.mul(f.add(vanish, f.one), zeta), zeta);                          // zeta^(n+2)
        FrEl cq = f.one;
        for (const auto& sp : split) {
            polys.push_back(sp);
            coeffs.push_back(f.mul(f.neg(vanish), cq));
            cq = f.mul(cq, z_n2);
        }
        void* d_lin = scope.alloc(PP);
        lincomb(polys, coeffs, d_lin, PP);
        // batch_poly = lin_poly + v w_0 + ... + v^5 w_4 + v^6 sigma_0 + ... + v^9 sigma_3 (:646-649)
        std::vector<std::pair<const void*, size_t>> bterms = {{d_lin, PP}};
        std::vector<FrEl> bcoef = {f.one};
        FrEl vp = v;
        for (int i = 0; i < 5; i++) { bterms.push_back({at(d_wp, i * WP), WP}); bcoef.push_back(vp); vp = f.mul(vp, v); }
        for (int i = 0; i < 4; i++) { bterms.push_back({at(sig_, i * n), n}); bcoef.push_back(vp); vp = f.mul(vp, v); }
        void* d_batch = scope.alloc(PP);
        lincomb(bterms, bcoef, d_batch, PP);
        void* d_wit = scope.alloc(2 * PP);
        check(plonk_poly_div_linear_dev(ctx_, d_batch, PP, zeta.data(), d_wit));                   // :651-666
        check(plonk_poly_div_linear_dev(ctx_, d_pp, PP, zeta_w.data(), at(d_wit, PP)));            // :672-688
        const std::vector<Point> open = commit_many({{d_wit, PP - 1}, {at(d_wit, PP), PP - 1}});   // :690-697
        proof.opening_proof = open[0];
        proof.shifted_opening_proof = open[1];
        return proof;
    }

  private:
    struct Scope {
        Prover& p;
        std::vector<void*> mine;
        explicit Scope(Prover& pr) : p(pr) {}
        ~Scope() { for (void* q : mine) plonk_dev_free(p.ctx_, q); }
        void* alloc(size_t n_fr) {
            void* q = nullptr;
            check(plonk_dev_alloc(p.ctx_, (n_fr ? n_fr : 1) * 32, &q));
            mine.push_back(q);
            return q;
        }
    };
    static const void* at(const void* base, size_t n_fr) { return (const char*)base + n_fr * 32; }
    static void* at(void* base, size_t n_fr) { return (char*)base + n_fr * 32; }
    void* alloc(size_t n_fr) {
        void* q = nullptr;
        check(plonk_dev_alloc(ctx_, (n_fr ? n_fr : 1) * 32, &q));
        bufs_.push_back(q);
        return q;
    }
    // domain.ifft (dispatcher2.rs:300-309, 345-346, 426): n evaluations -> n coefficients; the evaluations are not modified
    void interpolate(void* d_tmp, const void* d_evals, void* d_coeffs) {
        check(plonk_memcpy_d2d(ctx_, d_tmp, d_evals, n_ * 32));
        check(plonk_ntt_dev(ctx_, d_tmp, d_coeffs, n_, 1, 0));
    }
    Point to_point(const uint64_t* jac) const {
        Point P;
        P.xy.assign(2 * fq_limbs64(curve_), 0);
        int inf = 0;
        check(plonk_g1_to_affine(curve_, jac, P.xy.data(), &inf));
        P.inf = inf != 0;
        return P;
    }
    Point commit(const void* d_poly, size_t len) {      // commit_polynomial (dispatcher2.rs:835-893)
        std::vector<uint64_t> jac(3 * fq_limbs64(curve_));
        check(plonk_commit_dev(ctx_, d_poly, len, jac.data()));
        return to_point(jac.data());
    }
    // the independent commitments of a round as ONE Pippenger problem (plonk_commit_many_dev)
    std::vector<Point> commit_many(const std::vector<std::pair<const void*, size_t>>& items) {
        const size_t K = items.size(), J = 3 * fq_limbs64(curve_);
        std::vector<const void*> ptrs(K);
        std::vector<size_t> lens(K);
        for (size_t i = 0; i < K; i++) { ptrs[i] = items[i].first; lens[i] = items[i].second; }
        std::vector<uint64_t> jac(K * J);
        check(plonk_commit_many_dev(ctx_, K, ptrs.data(), lens.data(), 0, jac.data()));
        std::vector<Point> out;
        for (size_t i = 0; i < K; i++) out.push_back(to_point(&jac[i * J]));
        return out;
    }
    FrEl eval(const void* d_poly, size_t len, const FrEl& point) {                 // DensePolynomial::evaluate (:545-555)
        FrEl out{};
        check(plonk_poly_eval_dev(ctx_, d_poly, len, point.data(), out.data()));
        return out;
    }
    void lincomb(const std::vector<std::pair<const void*, size_t>>& terms, const std::vector<FrEl>& coeffs, void* d_out, size_t out_len) {
        const size_t K = terms.size();
        std::vector<const void*> ptrs(K);
        std::vector<size_t> lens(K);
        std::vector<uint64_t> cf(4 * K);
        for (size_t i = 0; i < K; i++) { ptrs[i] = terms[i].first; lens[i] = terms[i].second; std::memcpy(&cf[4 * i], coeffs[i].data(), 32); }
        check(plonk_poly_lincomb_dev(ctx_, K, ptrs.data(), lens.data(), cf.data(), d_out, out_len));
    }

    Worker& w_;
    plonk_ctx* ctx_;
    int curve_;
    FrField f_;
    int log_n_;
    size_t n_, m_;
    void* sel_ = nullptr;
    void* sig_ = nullptr;
    std::vector<FrEl> k_;
    std::vector<void*> bufs_;
    VerifyingKey vk_;
    bool have_vk_ = false;
};

}  // namespace plonk
