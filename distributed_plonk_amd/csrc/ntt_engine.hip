// ntt_engine.hip — host-side planner / launcher for the LDS-tiled NTT passes (ntt_kernels.hpp),
// twiddle-table setup, and the Fr matrix transpose (transpose.rs:413 equivalent).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "constants.h"
#include "ntt_kernels.hpp"
#include "plonk_internal.hpp"

static int ilog2(uint64_t x) { int l = 0; while (((uint64_t)1 << (l + 1)) <= x) l++; return l; }
static_assert(NTT_LOG_RMAX == 9, "NttTables::max_log_r defaults to NTT_LOG_RMAX");

const FrParams& fr_params(int curve) { return curve == PLONK_BN254 ? BN254_FR_PARAMS : BLS12_381_FR_PARAMS; }

// ---------------------------------------------------------------------------------------------- tables
static Fr host_pow2k(const Fr& a, int k, const FrParams& P) {   // a^(2^k)
    Fr r = a;
    for (int i = 0; i < k; i++) r = fp_sqr(r, P);
    return r;
}

// first * base^i for i < count, converted to constant form (c*2^261 mod p, 29-bit limbs)
static int upload_powers(F29** d_out, const Fr& base, size_t count, const Fr& first, const FrParams& P, hipStream_t stream) {
    std::vector<F29> h(count);
    Fr r261;
    for (int i = 0; i < 8; i++) r261.l[i] = P.one[i];
    for (int i = 0; i < 5; i++) r261 = fp_add(r261, r261, P);
    const Fr k_mont = fp_to_mont(r261, P);
    Fr acc = first;
    for (size_t i = 0; i < count; i++) {
        h[i] = f29_from_sat(fp_from_mont(fp_mul(acc, k_mont, P), P));
        acc = fp_mul(acc, base, P);
    }
    HIP_TRY(hipMalloc((void**)d_out, count * sizeof(F29)));
    HIP_TRY(hipMemcpyAsync(*d_out, h.data(), count * sizeof(F29), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return PLONK_OK;
}

int ntt_tables_create(NttTables& T, int curve, hipStream_t stream) {
    T.curve = curve;
    T.fp = fr_params(curve);
    T.fp29 = f29_make_params(T.fp);
    const FrParams& P = T.fp;
    const uint32_t* root_l = curve == PLONK_BN254 ? BN254_FR_TWO_ADIC_ROOT_MONT : BLS12_381_FR_TWO_ADIC_ROOT_MONT;
    const uint32_t* g_l = curve == PLONK_BN254 ? BN254_FR_GENERATOR_MONT : BLS12_381_FR_GENERATOR_MONT;
    const uint32_t* gi_l = curve == PLONK_BN254 ? BN254_FR_GENERATOR_INV_MONT : BLS12_381_FR_GENERATOR_INV_MONT;
    T.two_adicity = curve == PLONK_BN254 ? BN254_FR_TWO_ADICITY : BLS12_381_FR_TWO_ADICITY;
    T.lt = (T.two_adicity + 1) / 2;
    const Fr one = fp_one(P);
    Fr w[2];
    w[0] = fp_from_limbs<8>(root_l);
    w[1] = fp_inv(w[0], P);
    T.h_root[0] = w[0];
    T.h_root[1] = w[1];
    Fr g[2] = {fp_from_limbs<8>(g_l), fp_from_limbs<8>(gi_l)};
    const size_t nt = (size_t)1 << T.lt;
    for (int d = 0; d < 2; d++) {
        int rc;
        // Nmax = 2^two_adicity (= 2^(2*lt) for both curves)
        if ((rc = upload_powers(&T.tw_lo[d], w[d], nt, one, P, stream))) return rc;
        if ((rc = upload_powers(&T.tw_hi[d], host_pow2k(w[d], T.lt, P), nt, one, P, stream))) return rc;
        if ((rc = upload_powers(&T.tw_small[d], host_pow2k(w[d], T.two_adicity - NTT_LOG_RMAX, P),
                                (size_t)1 << (NTT_LOG_RMAX - 1), one, P, stream))) return rc;
        {   // the in-LDS twiddles once more, prepared for f29_mul_shoup
            const size_t cnt = (size_t)1 << (NTT_LOG_RMAX - 1);
            std::vector<F29S> h(cnt);
            const Fr base = host_pow2k(w[d], T.two_adicity - NTT_LOG_RMAX, P);
            Fr acc = one;
            for (size_t i = 0; i < cnt; i++) { h[i] = f29_shoup_from_mont256(acc, P); acc = fp_mul(acc, base, P); }
            HIP_TRY(hipMalloc((void**)&T.tw_shoup[d], cnt * sizeof(F29S)));
            HIP_TRY(hipMemcpyAsync(T.tw_shoup[d], h.data(), cnt * sizeof(F29S), hipMemcpyHostToDevice, stream));
            HIP_TRY(hipStreamSynchronize(stream));
        }
        if ((rc = upload_powers(&T.g_lo[d], g[d], nt, one, P, stream))) return rc;
        if ((rc = upload_powers(&T.g_hi[d], host_pow2k(g[d], T.lt, P), nt, one, P, stream))) return rc;
    }
    // Both fields (round 4): the butterflies' 4p-per-stage growth reaches < 36p on BN254 and < 38p on BLS12-381 (whose inputs arrive < 1.6p: the
    // inter-pass Montgomery product of a 36p value returns < 36p * p / 2^261 + p); what the multipliers need is x < 2^261 = 70p (BLS12-381) with
    // limbs < 2^31 — checked on the host build against Python integers up to 40p (tests/test_fp29_host.py), ntt_kernels.hpp has the chain.
    T.use_shoup = getenv("PLONK_NTT_NO_SHOUP") == nullptr;
    // 2^-k
    Fr two = fp_add(one, one, P), half = fp_inv(two, P);
    T.h_pow2_inv.resize(T.two_adicity + 1);
    Fr acc = one;
    for (int k = 0; k <= T.two_adicity; k++) { T.h_pow2_inv[k] = acc; acc = fp_mul(acc, half, P); }
    return PLONK_OK;
}

// Everything a context can rebuild on demand: inter-pass / output-factor planes, coset row tables, the first-pass tables of class-decomposed
// evaluations, the quotient kernel's 1/(x - 1) planes, power tables — not the root tables the domains were set up with.  -> bytes' worth of
// entries released (the planes' share is T.plane_bytes; the rest is small).
void ntt_tables_trim(NttTables& T) {
    for (auto& kv : T.tw_lo_scaled) (void)hipFree(kv.second);
    T.tw_lo_scaled.clear();
    for (auto& kv : T.planes) (void)hipFree(kv.second);
    T.planes.clear();
    for (auto& kv : T.rowtabs) (void)hipFree(kv.second);
    T.rowtabs.clear();
    for (auto& kv : T.shift_sets) { (void)hipFree(kv.second.planes); (void)hipFree(kv.second.rowtabs); (void)hipFree(kv.second.foldc); }
    T.shift_sets.clear();
    T.plane_bytes = 0;
    for (auto& kv : T.quot_inv_xm1) (void)hipFree(kv.second);
    T.quot_inv_xm1.clear();
    for (auto& kv : T.pow_tabs) (void)hipFree(kv.second);
    T.pow_tabs.clear();
    T.pow_order.clear();
}

void ntt_tables_destroy(NttTables& T) {
    for (int d = 0; d < 2; d++) {
        (void)hipFree(T.tw_shoup[d]); T.tw_shoup[d] = nullptr;
        (void)hipFree(T.tw_small[d]); (void)hipFree(T.tw_lo[d]); (void)hipFree(T.tw_hi[d]); (void)hipFree(T.g_lo[d]); (void)hipFree(T.g_hi[d]);
        T.tw_small[d] = T.tw_lo[d] = T.tw_hi[d] = T.g_lo[d] = T.g_hi[d] = nullptr;
    }
    for (auto& kv : T.tw_lo_scaled) (void)hipFree(kv.second);
    T.tw_lo_scaled.clear();
    for (auto& kv : T.planes) (void)hipFree(kv.second);
    T.planes.clear();
    for (auto& kv : T.rowtabs) (void)hipFree(kv.second);
    T.rowtabs.clear();
    for (auto& kv : T.shift_sets) { (void)hipFree(kv.second.planes); (void)hipFree(kv.second.rowtabs); (void)hipFree(kv.second.foldc); }
    T.shift_sets.clear();
    T.plane_bytes = 0;
    if (T.quot_x_lo) { (void)hipFree(T.quot_x_lo); T.quot_x_lo = nullptr; }
    for (auto& kv : T.quot_inv_xm1) (void)hipFree(kv.second);
    T.quot_inv_xm1.clear();
    for (auto& kv : T.pow_tabs) (void)hipFree(kv.second);
    T.pow_tabs.clear();
    T.pow_order.clear();
}

// w^-e * 2^-log_m for e < 2^lt (inverse transforms of size 2^log_m fold their 1/M here)
static int get_scaled_lo(NttTables& T, int log_m, F29** out, hipStream_t stream) {
    auto it = T.tw_lo_scaled.find(log_m);
    if (it != T.tw_lo_scaled.end()) { *out = it->second; return PLONK_OK; }
    F29* d = nullptr;
    int rc = upload_powers(&d, T.h_root[1], (size_t)1 << T.lt, T.h_pow2_inv[log_m], T.fp, stream);
    if (rc) return rc;
    T.tw_lo_scaled[log_m] = d;
    *out = d;
    return PLONK_OK;
}

// A transform that cannot get its factor plane still runs (two table gathers + an extra product per element: ~15 % slower); the
// switch is a performance cliff, so it is announced once per process instead of happening silently.
static void plane_fallback_notice(const char* why, size_t bytes) {
    static bool said = false;
    if (said) return;
    said = true;
    fprintf(stderr, "[plonk_hip] notice: no NTT factor plane of %zu MiB (%s): inter-pass twiddles are formed on the fly (slower)\n", bytes >> 20, why);
}

// Inter-pass factor plane (ntt_gen_plane_kernel) for pass `p` of a size-2^log_m transform, cached per
// (log_m, r_prev, r_p, direction, inverse-scale folded, coset folded).  Returns nullptr (no error) when the plane
// budget is exhausted — the pass then forms its factors on the fly.
static int get_plane(NttTables& T, int log_m, int log_rprev, int log_rp, int dir, bool fold_scale, bool fold_coset, int log_coset_mult,
                     Fr** out, hipStream_t stream) {
    *out = nullptr;
    const uint64_t key = ((uint64_t)log_m << 32) | ((uint64_t)log_rprev << 24) | ((uint64_t)log_rp << 16) | ((uint64_t)log_coset_mult << 8) |
                         ((uint64_t)dir << 2) | ((uint64_t)fold_scale << 1) | (uint64_t)fold_coset;
    auto it = T.planes.find(key);
    if (it != T.planes.end()) { *out = it->second; return PLONK_OK; }
    const uint64_t r_prev = (uint64_t)1 << log_rprev;
    const size_t bytes = r_prev * sizeof(Fr);
    if (T.plane_bytes + bytes > T.plane_budget) { plane_fallback_notice("the plane budget is exhausted", bytes); return PLONK_OK; }
    F29* lo = T.tw_lo[dir];
    if (fold_scale) {
        int rc = get_scaled_lo(T, log_m, &lo, stream);
        if (rc) return rc;
    }
    Fr* d = nullptr;
    if (hipMalloc((void**)&d, bytes) != hipSuccess) { (void)hipGetLastError(); plane_fallback_notice("hipMalloc failed", bytes); return PLONK_OK; }
    hipLaunchKernelGGL(ntt_gen_plane_kernel, dim3((uint32_t)((r_prev + 255) / 256)), dim3(256), 0, stream, d, r_prev, (uint64_t)1 << log_rp, lo,
                       T.tw_hi[dir], (uint32_t)T.lt, (uint32_t)(T.two_adicity - log_rprev), fold_coset ? T.g_lo[0] : (const F29*)nullptr,
                       fold_coset ? T.g_hi[0] : (const F29*)nullptr, (uint64_t)1 << log_coset_mult, T.fp29);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { (void)hipFree(d); return plonk_fail(PLONK_ERR_HIP, "ntt_gen_plane launch: %s", hipGetErrorString(e)); }
    T.planes[key] = d;
    T.plane_bytes += bytes;
    *out = d;
    return PLONK_OK;
}

// Output-factor plane for the distributed row pass (see ntt_gen_epi_plane_kernel), cached per (N, M, batch, q0, dir, coset)
static int get_epi_plane(NttTables& T, int log_N, int log_M, uint64_t batch, uint64_t q0, int dir, bool fold_coset, Fr** out, hipStream_t stream) {
    *out = nullptr;
    const uint64_t key = ((uint64_t)1 << 63) | ((uint64_t)log_N << 56) | ((uint64_t)log_M << 50) | ((uint64_t)ilog2(batch) << 44) | (q0 << 8) |
                         ((uint64_t)dir << 1) | (uint64_t)fold_coset;
    auto it = T.planes.find(key);
    if (it != T.planes.end()) { *out = it->second; return PLONK_OK; }
    const uint64_t M = (uint64_t)1 << log_M;
    const size_t bytes = M * batch * sizeof(Fr);
    if (T.plane_bytes + bytes > T.plane_budget) return PLONK_OK;
    Fr* d = nullptr;
    if (hipMalloc((void**)&d, bytes) != hipSuccess) { (void)hipGetLastError(); return PLONK_OK; }
    hipLaunchKernelGGL(ntt_gen_epi_plane_kernel, dim3((uint32_t)((M * batch + 255) / 256)), dim3(256), 0, stream, d, M, batch, q0, T.tw_lo[dir],
                       T.tw_hi[dir], (uint32_t)T.lt, (uint32_t)(T.two_adicity - log_N), fold_coset ? T.g_lo[0] : (const F29*)nullptr,
                       fold_coset ? T.g_hi[0] : (const F29*)nullptr, T.fp29);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { (void)hipFree(d); return plonk_fail(PLONK_ERR_HIP, "ntt_gen_epi_plane launch: %s", hipGetErrorString(e)); }
    T.planes[key] = d;
    T.plane_bytes += bytes;
    *out = d;
    return PLONK_OK;
}

// G[a] = g^(a * 2^log_r1), a < 2^log_w  (forward coset shift split as g^(a*r_1) * g^b; the g^b half lives in the plane)
static int get_rowtab(NttTables& T, int log_r1, int log_w, F29** out, hipStream_t stream) {
    const uint64_t key = ((uint64_t)log_r1 << 8) | (uint64_t)log_w;
    auto it = T.rowtabs.find(key);
    if (it != T.rowtabs.end()) { *out = it->second; return PLONK_OK; }
    const uint32_t* g_l = T.curve == PLONK_BN254 ? BN254_FR_GENERATOR_MONT : BLS12_381_FR_GENERATOR_MONT;
    Fr base = fp_from_limbs<8>(g_l);
    for (int i = 0; i < log_r1; i++) base = fp_sqr(base, T.fp);         // g^(r_1)
    F29* d = nullptr;
    int rc = upload_powers(&d, base, (size_t)1 << log_w, fp_one(T.fp), T.fp, stream);
    if (rc) return rc;
    T.rowtabs[key] = d;
    *out = d;
    return PLONK_OK;
}

// c * base^i, i < count, appended to `h` in constant form
static void host_powers_const(std::vector<F29>& h, const Fr& base, size_t count, const FrParams& P) {
    Fr r261;
    for (int i = 0; i < 8; i++) r261.l[i] = P.one[i];
    for (int i = 0; i < 5; i++) r261 = fp_add(r261, r261, P);
    const Fr k_mont = fp_to_mont(r261, P);
    Fr acc = fp_one(P);
    for (size_t i = 0; i < count; i++) {
        h.push_back(f29_from_sat(fp_from_mont(fp_mul(acc, k_mont, P), P)));
        acc = fp_mul(acc, base, P);
    }
}

// First-pass tables for the evaluation of one coefficient vector on the B cosets h_q * <w_M>, h_q = h * w_(M*B)^q:
// row tables h_q^(a*r_1), fold constants (h_q^M)^u and — for transforms of more than one pass — the planes w_M^(b*i) * h_q^b.
// Cached per (h, log M, log B, first pass width); planes count against the plane budget (older sets are dropped to make room;
// running out of memory is an ERROR here, not a silent slow path: there is no plane-less variant of this transform).
static int get_shift_set(NttTables& T, const Fr& h, int L, uint64_t B, int w0, int NP, const NttTables::ShiftSet** out, hipStream_t stream) {
    const FrParams& P = T.fp;
    const int logB = ilog2(B);
    std::string key((const char*)h.l, 32);
    key.push_back((char)L); key.push_back((char)logB); key.push_back((char)w0);
    auto it = T.shift_sets.find(key);
    if (it != T.shift_sets.end()) { *out = &it->second; return PLONK_OK; }
    if (L + logB > T.two_adicity) return plonk_fail(PLONK_ERR_DOMAIN, "coset classes: 2^%d points exceed the two-adicity", L + logB);
    const uint64_t M = (uint64_t)1 << L, R1 = (uint64_t)1 << w0, r1 = M >> w0;
    NttTables::ShiftSet set;
    set.bytes = NP > 1 ? (size_t)(B * M) * sizeof(Fr) : 0;
    if (T.plane_bytes + set.bytes > T.plane_budget && !T.shift_sets.empty()) {        // make room: drop the other shift sets
        HIP_TRY(hipStreamSynchronize(stream));
        for (auto& kv : T.shift_sets) {
            (void)hipFree(kv.second.planes); (void)hipFree(kv.second.rowtabs); (void)hipFree(kv.second.foldc);
            T.plane_bytes -= kv.second.bytes;
        }
        T.shift_sets.clear();
    }
    if (T.plane_bytes + set.bytes > T.plane_budget)
        return plonk_fail(PLONK_ERR_HIP, "coset classes: %zu bytes of first-pass planes exceed the plane budget (%zu of %zu in use)", set.bytes,
                          T.plane_bytes, T.plane_budget);
    const Fr wB = host_pow2k(T.h_root[0], T.two_adicity - (L + logB), P);          // primitive (M*B)-th root of unity
    std::vector<F29> h_row, h_fold;
    h_row.reserve(B * R1); h_fold.reserve(B * NTT_MAX_FOLD);
    std::vector<Fr> hq(B);
    Fr acc = h;
    for (uint64_t q = 0; q < B; q++) { hq[q] = acc; acc = fp_mul(acc, wB, P); }
    for (uint64_t q = 0; q < B; q++) {
        host_powers_const(h_row, host_pow2k(hq[q], L - w0, P), R1, P);             // (h_q^r_1)^a
        host_powers_const(h_fold, host_pow2k(hq[q], L, P), NTT_MAX_FOLD, P);                  // (h_q^M)^u
    }
    struct Guard {                 // frees whatever has been allocated unless the set is handed over
        NttTables::ShiftSet* s; F29* extra = nullptr; bool armed = true;
        ~Guard() { if (armed) { (void)hipFree(s->planes); (void)hipFree(s->rowtabs); (void)hipFree(s->foldc); } (void)hipFree(extra); }
    } guard{&set};
    HIP_TRY(hipMalloc((void**)&set.rowtabs, h_row.size() * sizeof(F29)));
    HIP_TRY(hipMalloc((void**)&set.foldc, h_fold.size() * sizeof(F29)));
    HIP_TRY(hipMemcpyAsync(set.rowtabs, h_row.data(), h_row.size() * sizeof(F29), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(set.foldc, h_fold.data(), h_fold.size() * sizeof(F29), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (NP > 1) {
        if (hipMalloc((void**)&set.planes, set.bytes) != hipSuccess) {
            (void)hipGetLastError();
            set.planes = nullptr;
            return plonk_fail(PLONK_ERR_HIP, "coset classes: out of memory for %zu bytes of first-pass planes", set.bytes);
        }
        const size_t n_hi = (size_t)std::max<uint64_t>(1, r1 >> 10);
        std::vector<F29> h_pw;
        h_pw.reserve(B * (1024 + n_hi));
        for (uint64_t q = 0; q < B; q++) {
            host_powers_const(h_pw, hq[q], 1024, P);
            host_powers_const(h_pw, host_pow2k(hq[q], 10, P), n_hi, P);
        }
        F29* d_pw = nullptr;
        HIP_TRY(hipMalloc((void**)&d_pw, h_pw.size() * sizeof(F29)));
        guard.extra = d_pw;
        HIP_TRY(hipMemcpyAsync(d_pw, h_pw.data(), h_pw.size() * sizeof(F29), hipMemcpyHostToDevice, stream));
        for (uint64_t q = 0; q < B; q++) {
            const F29* lo = d_pw + q * (1024 + n_hi);
            hipLaunchKernelGGL(ntt_gen_shift_plane_kernel, dim3((uint32_t)((M + 255) / 256)), dim3(256), 0, stream, set.planes + q * M, M, r1, T.tw_lo[0],
                               T.tw_hi[0], (uint32_t)T.lt, (uint32_t)(T.two_adicity - L), lo, lo + 1024, T.fp29);
        }
        hipError_t e = hipGetLastError();
        hipError_t e2 = hipStreamSynchronize(stream);
        if (e != hipSuccess || e2 != hipSuccess) return plonk_fail(PLONK_ERR_HIP, "ntt_gen_shift_plane: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    }
    guard.armed = false;
    T.plane_bytes += set.bytes;
    auto ins = T.shift_sets.emplace(key, set);
    *out = &ins.first->second;
    return PLONK_OK;
}

// ---------------------------------------------------------------------------------------------- plan
std::vector<int> ntt_plan_widths(int log_m, int max_log_r) {
    std::vector<int> w;
    const int mx = std::max(3, std::min(max_log_r, NTT_LOG_RMAX));
    if (log_m <= mx) { w.push_back(log_m); return w; }
    int P = (log_m + mx - 1) / mx;
    int base = log_m / P, rem = log_m % P;
    for (int i = 0; i < P; i++) w.push_back(base + (i < rem ? 1 : 0));
    return w;
}


static int pref_log_t(int log_r) {
    static const char* ov[4] = {getenv("PLONK_NTT_LOGT6"), getenv("PLONK_NTT_LOGT7"), getenv("PLONK_NTT_LOGT8"), getenv("PLONK_NTT_LOGT9")};   // tuning overrides (experiments)
    if (log_r >= 6 && ov[std::min(log_r, 9) - 6]) return atoi(ov[std::min(log_r, 9) - 6]);
    if (log_r >= 9) return 3;  // 4096-element tile: 144 KiB of the CU's 160 KiB LDS
    if (log_r < 3) return 8;   // EPT = R there: one lane per column
    // 2^6- and 2^7-row passes (the 7 + 7 + 6 / 7 + 7 + 8 plans of 2^20 ... 2^22-point transforms — configs[1], configs[3] — and the row / column
    // transforms of the distributed 2-D NTT) take 8-column tiles as well (18 / 36 KiB, eight / four workgroups per CU): that is the shape the bank
    // swizzle and the precomputed-quotient butterflies are instantiated for — with 32 / 16 columns these passes ran the generic kernel.  Round 4,
    // same box: 8n coset FFT 2^19 0.65 -> 0.58 ms, 2^20 1.11 -> 1.03, 2^21 1.97 -> 1.79, 2^22 3.85 -> 3.68 (BLS12-381: 3.87 -> 3.66), dense 2^12
    // 0.075 -> 0.059, 2^19 0.215 -> 0.188; 2^23 and above unchanged; 2^20 step -3.5 % (profiles/r04_small_plans_experiment.txt).
    if (log_r == 6 || log_r == 7) return 3;
    // 2^5-row passes only exist in the 5 + 5 / 6 + 5 plans of 2^10- and 2^11-point transforms — the zero-padding-aware row pass of the 8-rank
    // 2-D transform is 8 class transforms of 2^11 points per row (r = 2^13, c = 2^14, n + 3 coefficients in 8n).  Round 6: one wavefront per
    // workgroup on an 8-column tile (9 KiB of LDS, barriers are free) instead of the generic kernel on 64 columns (VERDICT r5 weak 3:
    // 25 launches per simulated step at 3.1-3.5 ms against 1.05 ms for the tuned 2^7-row passes beside them).  PLONK_NTT_LOGT5 overrides.
    if (log_r == 5) { static const char* ov5 = getenv("PLONK_NTT_LOGT5"); return ov5 ? atoi(ov5) : 3; }
    return 11 - log_r;         // 2048-element tiles (72 KiB: two workgroups per CU)
}

bool ntt_single_pass_inplace_ok(const NttTables& T, const NttCall& c) {
    return ntt_plan_widths(c.log_m, T.max_log_r).size() == 1 && c.layout == NTT_CONTIGUOUS && c.out_layout == NTT_CONTIGUOUS && c.split_log < 0;
}

static TwoLevelScale make_scale(const NttTables& T, const ScaleSpec& s, uint64_t q_offset) {
    TwoLevelScale o;
    memset(&o, 0, sizeof o);
    if (s.kind == 0) return o;
    o.enabled = 1;
    o.lt = T.lt;
    uint64_t aq = s.aq, a0 = s.a0 + s.aq * q_offset, bq = s.bq, b0 = s.b0 + s.bq * q_offset;
    if (s.kind == 1 || s.kind == 2) {
        o.lo = T.g_lo[s.kind - 1];
        o.hi = T.g_hi[s.kind - 1];
    } else {
        const int d = s.kind - 3;
        o.lo = T.tw_lo[d];
        o.hi = T.tw_hi[d];
        const int sh = T.two_adicity - s.log_order;
        aq <<= sh; a0 <<= sh; bq <<= sh; b0 <<= sh;
    }
    o.aq = aq; o.a0 = a0; o.bq = bq; o.b0 = b0;
    return o;
}

// elements per lane: 4 (radix-4 steps; the measured choice) — PLONK_NTT_EPT = 8 / 2 selects the other instantiations for experiments.
// Read once per process (a thread-safe function-local static), never changed afterwards.
static int ntt_ept() {
    static const int ept = [] {
        const char* e = getenv("PLONK_NTT_EPT");
        const int v = e ? atoi(e) : 4;
        return (v == 4 || v == 2) ? v : (e ? 8 : 4);
    }();
    return ept;
}

template <int LOG_R, int EPT, bool SWZ = false, bool SHOUP = false>
static hipError_t launch_one_e(const NttPassParams& P, uint64_t grid, uint32_t threads, size_t lds, hipStream_t stream) {
    static DeviceOnce attr;
    hipError_t e = attr.run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt_pass_kernel<LOG_R, EPT, SWZ, SHOUP>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((ntt_pass_kernel<LOG_R, EPT, SWZ, SHOUP>), dim3((uint32_t)grid), dim3(threads), lds, stream, P);
    return hipGetLastError();
}
static bool swizzle_on() {
    static const bool on = getenv("PLONK_NTT_NO_SWIZZLE") == nullptr;
    return on;
}
template <int LOG_R>
static hipError_t launch_one(const NttPassParams& P, uint64_t grid, uint32_t threads, size_t lds, hipStream_t stream) {
    // the bank swizzle (ntt_kernels.hpp: sw_fold) is built for the production tile shape: 8 columns, 4 elements per lane (rows >= 2^5: every
    // width a multi-pass plan can contain, ntt_plan_widths)
    if constexpr (LOG_R >= 5) {
        if (ntt_ept() == 4 && P.log_t == 3 && P.tile_pitch == 8 && swizzle_on()) {
            if (P.tw_shoup != nullptr) return launch_one_e<LOG_R, 4, true, true>(P, grid, threads, lds, stream);
            return launch_one_e<LOG_R, 4, true>(P, grid, threads, lds, stream);
        }
    }
    if (ntt_ept() == 4) return launch_one_e<LOG_R, 4>(P, grid, threads, lds, stream);
    if (ntt_ept() == 2) return launch_one_e<LOG_R, 2>(P, grid, threads, lds, stream);
    return launch_one_e<LOG_R, 8>(P, grid, threads, lds, stream);
}

static hipError_t launch_pass(int log_r, const NttPassParams& P, uint64_t grid, uint32_t threads, size_t lds, hipStream_t s) {
    switch (log_r) {
        case 1: return launch_one<1>(P, grid, threads, lds, s);
        case 2: return launch_one<2>(P, grid, threads, lds, s);
        case 3: return launch_one<3>(P, grid, threads, lds, s);
        case 4: return launch_one<4>(P, grid, threads, lds, s);
        case 5: return launch_one<5>(P, grid, threads, lds, s);
        case 6: return launch_one<6>(P, grid, threads, lds, s);
        case 7: return launch_one<7>(P, grid, threads, lds, s);
        case 8: return launch_one<8>(P, grid, threads, lds, s);
        case 9: return launch_one<9>(P, grid, threads, lds, s);
    }
    return hipErrorInvalidValue;
}

static bool planes_on_global() {
    static const bool on = getenv("PLONK_NTT_NO_PLANES") == nullptr;
    return on;
}

int ntt_run(NttTables& T, const NttCall& c, hipStream_t stream) {
    const int L = c.log_m;
    if (L < 1 || L > T.two_adicity) return plonk_fail(PLONK_ERR_DOMAIN, "ntt_run: log size %d outside [1,%d]", L, T.two_adicity);
    if (c.batch == 0 || (c.batch & (c.batch - 1))) return plonk_fail(PLONK_ERR_ARG, "ntt_run: batch must be a power of two");
    const uint64_t M = (uint64_t)1 << L, Bt = c.batch;
    const std::vector<int> widths = ntt_plan_widths(L, T.max_log_r);
    const int NP = (int)widths.size();
    if (NP > NTT_MAX_PASSES) return plonk_fail(PLONK_ERR_ARG, "ntt_run: too many passes");
    const int dir = c.inverse ? 1 : 0;
    const bool interleaved = c.layout == NTT_INTERLEAVED;
    if (NP > 1 && !c.shared_in && (const void*)c.in == (const void*)c.out) return plonk_fail(PLONK_ERR_ARG, "ntt_run: in == out needs a single pass");
    const NttTables::ShiftSet* sset = nullptr;
    if (c.shared_in) {
        const bool epi_ok = c.epi.kind == 0 || ((c.epi.kind == 3 || c.epi.kind == 4) && c.epi.bq == 1 && c.epi.b0 == 0 && c.epi.aq == 0 && c.epi.a0 == 0);
        if (interleaved || c.inverse || c.pro.kind || !epi_ok || c.in_len == 0 || c.in_len > NTT_MAX_FOLD * M || c.in_rows == 0 || Bt % c.in_rows ||
            (c.row_coset_const && c.epi.kind == 0))
            return plonk_fail(PLONK_ERR_ARG, "ntt_run: shared-input evaluation is forward, contiguous, 1 <= len <= %dM, batch = rows * classes", NTT_MAX_FOLD);
        if (NP > 1 && (c.work == nullptr || (const void*)c.work == (const void*)c.out || (const void*)c.work == (const void*)c.in))
            return plonk_fail(PLONK_ERR_ARG, "ntt_run: shared-input evaluation needs a separate work buffer");
        if (NP == 1 && (const void*)c.in == (const void*)c.out) return plonk_fail(PLONK_ERR_ARG, "ntt_run: shared-input evaluation cannot run in place");
        int rc = get_shift_set(T, c.shift, L, Bt / c.in_rows, widths[0], NP, &sset, stream);
        if (rc) return rc;
    }
    const uint64_t n_cls = c.shared_in ? Bt / c.in_rows : 1;                 // classes per row (shared-input mode)
    const int cls_log = ilog2(n_cls);
    Fr* const inplace = c.shared_in ? c.work : const_cast<Fr*>(c.in);      // where the non-last passes leave their output
    if (NP == 1 && (const void*)c.in == (const void*)c.out && !ntt_single_pass_inplace_ok(T, c))
        return plonk_fail(PLONK_ERR_ARG, "ntt_run: in-place only for contiguous single pass");

    // forward coset shift of a whole contiguous vector: x[n] * g^n with n = a*r_1 + b splits into a per-row table
    // (g^(a*r_1), applied at load) and g^b folded into the first inter-pass plane
    // Forward coset shifts handled without per-element exponent arithmetic:  g^(pos*B0 + AQ*(q+q0)),  B0 a power of two,
    // pos = a*r_1 + b   =>   row table g^(B0*r_1*a) at load,  g^(B0*b) folded into the first plane,  and the per-array
    // constant g^(q+q0) (AQ = 1: the distributed row pass, worker.rs:75-80) folded into the output-factor plane.
    const bool pro_pow2 = c.pro.b0 != 0 && (c.pro.b0 & (c.pro.b0 - 1)) == 0;
    const bool simple_coset = c.pro.kind == 1 && c.pro.a0 == 0 && c.pro.bq == 0 && pro_pow2 && c.pro.aq <= 1;
    const int log_b0 = simple_coset ? ilog2(c.pro.b0) : 0;
    // the distributed row pass's output factor w_N^(+-(q+q0)*k)
    const bool row_twiddle = (c.epi.kind == 3 || c.epi.kind == 4) && c.epi.bq == 1 && c.epi.b0 == 0 && c.epi.aq == 0 && c.epi.a0 == 0;
    const bool need_q_const = (simple_coset && c.pro.aq == 1) || (c.shared_in && c.row_coset_const);      // only foldable into an output-factor plane
    Fr* epi_plane = nullptr;
    if (c.shared_in) {
        if (row_twiddle) {          // the plane covers whole rows: 2^(L + cls_log) natural-order outputs per row, in_rows rows
            int rc = get_epi_plane(T, c.epi.log_order, L + cls_log, c.in_rows, c.q_offset, c.epi.kind - 3, need_q_const, &epi_plane, stream);
            if (rc) return rc;
            if (epi_plane == nullptr) return plonk_fail(PLONK_ERR_HIP, "ntt_run: no room for the output-factor plane of the zero-padded row pass");
        }
    } else if (planes_on_global() && row_twiddle && !interleaved && (!need_q_const || NP >= 2)) {
        int rc = get_epi_plane(T, c.epi.log_order, L, Bt, c.q_offset, c.epi.kind - 3, need_q_const && NP >= 2, &epi_plane, stream);
        if (rc) return rc;
    }
    const bool coset_foldable = simple_coset && (!need_q_const || epi_plane != nullptr);
    bool coset_folded = false;
    const bool planes_on = planes_on_global();
    uint64_t r_prev = M;
    for (int p = 0; p < NP; p++) {
        const int w = widths[p];
        const uint64_t R = (uint64_t)1 << w;
        const uint64_t r_p = r_prev >> w;
        const bool last = (p == NP - 1);
        NttPassParams P;
        memset(&P, 0, sizeof P);
        P.fp = T.fp29;
        P.tw_small = T.tw_small[dir];
        P.tw_shoup = T.use_shoup ? T.tw_shoup[dir] : nullptr;
        P.tw_lo = T.tw_lo[dir];
        P.tw_hi = T.tw_hi[dir];
        P.tw_lt = T.lt;
        P.split_log = -1;
        P.is_last = last ? 1 : 0;
        P.in = (p == 0) ? c.in : inplace;
        P.cls_log = (uint32_t)cls_log;
        if (sset != nullptr && p == 0) {
            P.in_row_pitch = c.in_pitch;
            P.in_len = c.in_len;
            P.fold_m = M;
            P.nfold = (uint32_t)std::min<uint64_t>(NTT_MAX_FOLD, (c.in_len + M - 1) / M);
            P.fold_c = sset->foldc;
            P.pro_rowtab = sset->rowtabs;
            P.rowtab_qstride = (uint64_t)1 << widths[0];
        }
        uint64_t grid = 0, avail = 1;
        if (!last) {
            P.out = inplace;
            P.tw_shift = T.two_adicity - ilog2(r_prev);
            if (p == 0 && c.inverse) {
                F29* scaled = nullptr;
                int rc = get_scaled_lo(T, L, &scaled, stream);
                if (rc) return rc;
                P.tw_lo = scaled;
            }
            if (sset != nullptr && p == 0) {
                P.tw_plane = sset->planes;
                P.plane_qstride = M;
                P.plane_rp = r_p;
            } else if (planes_on) {
                const bool want_coset = (p == 0 && coset_foldable && !interleaved);
                Fr* plane = nullptr;
                int rc = get_plane(T, L, ilog2(r_prev), ilog2(r_p), dir, p == 0 && c.inverse, want_coset, want_coset ? log_b0 : 0, &plane, stream);
                if (rc) return rc;
                if (plane == nullptr && want_coset) {       // no room for the folded plane: try the plain one, keep the shift in the prologue
                    rc = get_plane(T, L, ilog2(r_prev), ilog2(r_p), dir, p == 0 && c.inverse, false, 0, &plane, stream);
                    if (rc) return rc;
                } else if (plane != nullptr && want_coset) {
                    coset_folded = true;
                    int rc2 = get_rowtab(T, log_b0 + ilog2(r_p), w, const_cast<F29**>(&P.pro_rowtab), stream);
                    if (rc2) return rc2;
                }
                P.tw_plane = plane;
                P.plane_rp = r_p;
            }
            avail = interleaved ? Bt : r_p;
            int lt = std::min(pref_log_t(w), ilog2(avail));
            const uint64_t Tt = (uint64_t)1 << lt;
            P.log_t = lt;
            if (!interleaved) {
                P.n0 = r_p / Tt; P.n1 = M / r_prev;
                P.ls0 = Tt; P.ls1 = r_prev; P.ls2 = M; P.l_astride = r_p; P.l_tstride = 1;
                P.bs0 = Tt; P.bs1 = 0; P.tb = 1;
                P.qs0 = 0; P.qs1 = 0; P.qs2 = 1; P.tq = 0;
                P.pa = r_p; P.ps0 = Tt; P.ps1 = 0; P.pt = 1;
                grid = Bt * P.n1 * P.n0;
            } else {
                P.n0 = Bt / Tt; P.n1 = r_p;
                P.ls0 = Tt; P.ls1 = Bt; P.ls2 = r_prev * Bt; P.l_astride = r_p * Bt; P.l_tstride = 1;
                P.bs0 = 0; P.bs1 = 1; P.tb = 0;
                P.qs0 = Tt; P.qs1 = 0; P.qs2 = 0; P.tq = 1;
                P.pa = r_p; P.ps0 = 0; P.ps1 = 1; P.pt = 0;
                grid = (M / r_prev) * P.n1 * P.n0;
            }
            P.ss0 = P.ls0; P.ss1 = P.ls1; P.ss2 = P.ls2; P.s_istride = P.l_astride; P.s_tstride = 1;
            P.tile_pitch = (uint32_t)Tt;
        } else {
            P.out = c.out;
            P.kstride = M >> w;
            if (sset != nullptr && NP >= 2 && n_cls >= 2) {
                // the tile's columns are the ARRAYS (cosets): contiguous runs in, and for every output index k the T arrays' values
                // leave as one T*32-byte piece of the interleaved (natural-order) result
                avail = Bt;
                int lt = std::min(pref_log_t(w), ilog2(avail));
                const uint64_t Tt = (uint64_t)1 << lt;
                P.log_t = lt;
                P.n0 = Bt / Tt; P.n1 = M >> w;
                P.ls0 = Tt * M; P.ls1 = R; P.ls2 = 0; P.l_astride = 1; P.l_tstride = M; P.load_a_fast = 1;
                P.qs0 = Tt; P.tq = 1;
                P.ms0 = 0; P.tk = 0;
                P.rev_ndig = NP - 1;
                for (int d = 0; d < NP - 1; d++) P.rev_w[d] = widths[d];
                P.rev_shift0 = 0;
                grid = P.n0 * P.n1;
            } else if (!interleaved && NP >= 2) {
                const uint64_t R1 = (uint64_t)1 << widths[0], r1 = M >> widths[0];
                avail = R1;
                int lt = std::min(pref_log_t(w), ilog2(avail));
                const uint64_t Tt = (uint64_t)1 << lt;
                P.log_t = lt;
                P.n0 = R1 / Tt; P.n1 = (M >> w) / R1;
                P.ls0 = Tt * r1; P.ls1 = R; P.ls2 = M; P.l_astride = 1; P.l_tstride = r1; P.load_a_fast = 1;
                P.qs2 = 1; P.tq = 0;
                P.ms0 = Tt; P.tk = 1;
                P.rev_ndig = NP - 2;
                for (int d = 0; d < NP - 2; d++) P.rev_w[d] = widths[1 + d];
                P.rev_shift0 = widths[0];
                grid = Bt * P.n1 * P.n0;
            } else if (!interleaved) {       // single pass, contiguous: tile over arrays
                avail = Bt;
                int lt = std::min(pref_log_t(w), ilog2(avail));
                const uint64_t Tt = (uint64_t)1 << lt;
                P.log_t = lt;
                P.n0 = Bt / Tt; P.n1 = 1;
                P.ls0 = Tt * M; P.l_astride = 1; P.l_tstride = M; P.load_a_fast = 1;
                P.qs0 = Tt; P.tq = 1;
                P.ms0 = 0; P.tk = 0; P.rev_ndig = 0;
                P.pa = 1;
                grid = P.n0;
            } else {
                avail = Bt;
                int lt = std::min(pref_log_t(w), ilog2(avail));
                const uint64_t Tt = (uint64_t)1 << lt;
                P.log_t = lt;
                P.n0 = Bt / Tt; P.n1 = M >> w;
                P.ls0 = Tt; P.ls1 = R * Bt; P.l_astride = Bt; P.l_tstride = 1; P.load_a_fast = 0;
                P.qs0 = Tt; P.tq = 1;
                P.ms0 = 0; P.tk = 0;
                P.rev_ndig = NP - 1;
                for (int d = 0; d < NP - 1; d++) P.rev_w[d] = widths[d];
                P.rev_shift0 = 0;
                P.pa = 1;
                grid = P.n0 * P.n1;
            }
            if (sset != nullptr) { P.oq = M * n_cls; P.ok = 1; }         // rows contiguous, a row's classes interleaved (kernel: k' = k << cls_log | class)
            else if (c.out_layout == NTT_CONTIGUOUS) { P.oq = M; P.ok = 1; } else { P.oq = 1; P.ok = Bt; }
            P.split_log = c.split_log;
            P.split_blk = c.split_blk;
            if (c.inverse && NP == 1) { P.scale_const_enabled = 1; P.scale_const = f29_const_from_mont256(T.h_pow2_inv[L], T.fp); }
            if (epi_plane != nullptr && (!need_q_const || coset_folded || sset != nullptr)) {
                P.epi_plane = epi_plane;          // q here is the LOCAL array index: the plane was generated with q0 added
                P.epi_qstride = M * n_cls;
            } else {
                P.epi = make_scale(T, c.epi, c.q_offset);
            }
            P.tile_pitch = (uint32_t)((uint64_t)1 << P.log_t);
        }
        if (p == 0 && !coset_folded) P.pro = make_scale(T, c.pro, c.q_offset);
        const uint64_t Tt = (uint64_t)1 << P.log_t;
        const uint32_t ept = (R >= (uint64_t)ntt_ept()) ? (uint32_t)ntt_ept() : (uint32_t)R;
        const uint32_t threads = (uint32_t)(R * Tt / ept);
        if (threads > 1024) return plonk_fail(PLONK_ERR_ARG, "ntt_run: tile of %llu x %llu needs %u lanes", (unsigned long long)R, (unsigned long long)Tt, threads);
        const size_t lds = (size_t)9 * R * P.tile_pitch * 4 + std::max<size_t>(R / 2, 1) * 36;
        if (grid == 0 || grid > 0x7fffffffull) return plonk_fail(PLONK_ERR_ARG, "ntt_run: grid %llu out of range", (unsigned long long)grid);
        static const char* const kNames[10] = {"", "ntt_pass_kernel<1>", "ntt_pass_kernel<2>", "ntt_pass_kernel<3>", "ntt_pass_kernel<4>",
                                               "ntt_pass_kernel<5>", "ntt_pass_kernel<6>", "ntt_pass_kernel<7>", "ntt_pass_kernel<8>",
                                               "ntt_pass_kernel<9>"};
        hipError_t e;
        {
            ProfScope ps_all("ntt_pass_kernel", stream);
            ProfScope ps_w(kNames[w], stream);
            e = launch_pass(w, P, grid, threads, lds, stream);
        }
        if (e != hipSuccess) return plonk_fail(PLONK_ERR_HIP, "ntt pass launch (log_r=%d, grid=%llu, threads=%u, lds=%zu): %s", w,
                                               (unsigned long long)grid, threads, lds, hipGetErrorString(e));
        r_prev = r_p;
    }
    return PLONK_OK;
}

// ---------------------------------------------------------------------------------------------- transpose
// out[c][r] = in[r][c] for a rows x cols matrix of Fr.  16x16-element tiles staged through LDS so
// that both the load and the store touch 512-byte contiguous pieces.
__global__ void __launch_bounds__(256) transpose_fr_kernel(const Fr* __restrict__ in, Fr* __restrict__ out, uint64_t rows, uint64_t cols) {
    __shared__ uint4 tile[16][16 * 2 + 1];
    const uint64_t tiles_c = (cols + 15) / 16;
    const uint64_t tr = blockIdx.x / tiles_c, tc = blockIdx.x % tiles_c;
    const uint32_t ly = threadIdx.x / 16, lx = threadIdx.x % 16;
    uint64_t r = tr * 16 + ly, cc = tc * 16 + lx;
    if (r < rows && cc < cols) {
        const uint4* src = reinterpret_cast<const uint4*>(in + r * cols + cc);
        tile[ly][2 * lx] = src[0];
        tile[ly][2 * lx + 1] = src[1];
    }
    __syncthreads();
    uint64_t orow = tc * 16 + ly, ocol = tr * 16 + lx;     // out is cols x rows
    if (orow < cols && ocol < rows) {
        uint4* dst = reinterpret_cast<uint4*>(out + orow * rows + ocol);
        dst[0] = tile[lx][2 * ly];
        dst[1] = tile[lx][2 * ly + 1];
    }
}

int transpose_fr(const Fr* in, Fr* out, uint64_t rows, uint64_t cols, hipStream_t stream) {
    if (rows == 0 || cols == 0) return PLONK_OK;
    if (in == out) return plonk_fail(PLONK_ERR_ARG, "transpose_fr: in == out");
    const uint64_t grid = ((rows + 15) / 16) * ((cols + 15) / 16);
    if (grid > 0x7fffffffull) return plonk_fail(PLONK_ERR_ARG, "transpose_fr: too large");
    hipLaunchKernelGGL(transpose_fr_kernel, dim3((uint32_t)grid), dim3(256), 0, stream, in, out, rows, cols);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return plonk_fail(PLONK_ERR_HIP, "transpose launch: %s", hipGetErrorString(e));
    return PLONK_OK;
}
