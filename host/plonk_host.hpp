// plonk_host.hpp — the host side of the hot path in C++, on nothing but the C ABI of include/plonk_hip.h.
//
// The reference's host side is Rust (no toolchain in this image: ffi/plonk_hip.rs is its binding as source); this header is the
// same orchestration as compiled code, so the drop-in boundary is exercised without Python:
//   plonk::Worker      one GPU context behind the PlonkSlave surface (reference `State` + `impl plonk_slave::Server`,
//                      /root/reference/src/worker.rs:42-59, 125-439)
//   plonk::Dispatcher  S in-process workers behind the dispatcher's call sequence: `Prover::fft`
//                      (src/dispatcher2.rs:732-787: resize, decimate, one fft1 per row, fft2_prepare on all workers, fft2, undecimate),
//                      the sharded MSM (src/dispatcher.rs:218-238) and `commit_polynomial` (src/dispatcher2.rs:835-893)
// Field elements are 4 x u64 (Montgomery, utils.rs:27-43); errors become exceptions carrying plonk_last_error().
// tests/host_cpp/host_check.cpp drives it against the oracle on an MI355X.
#pragma once
#include <cstdint>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "plonk_hip.h"

namespace plonk {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc) {
    if (rc != PLONK_OK) throw Error(rc, plonk_last_error());
}

using Fr = uint64_t[4];
inline size_t fq_limbs64(int curve) { return curve == PLONK_BN254 ? 4 : 6; }

// worker.rs:143-144, dispatcher2.rs:744-745
inline void split_rc(size_t n, size_t* r, size_t* c) {
    int log_n = 0;
    while (((size_t)1 << log_n) < n) log_n++;
    if (((size_t)1 << log_n) != n) throw Error(PLONK_ERR_DOMAIN, "domain size must be a power of two");
    *r = (size_t)1 << (log_n >> 1);
    *c = n / *r;
}
// dispatcher2.rs:272-291
inline std::vector<plonk_fft_workload> make_fft_workloads(size_t n, size_t s) {
    size_t r, c;
    split_rc(n, &r, &c);
    std::vector<plonk_fft_workload> w(s);
    for (size_t i = 0; i < s; i++) w[i] = {i * r / s, (i + 1) * r / s, i * c / s, (i + 1) * c / s};
    return w;
}
// dispatcher.rs:223-226
inline std::vector<plonk_msm_workload> make_msm_workloads(size_t n, size_t s) {
    std::vector<plonk_msm_workload> w(s);
    for (size_t i = 0; i < s; i++) w[i] = {i * n / s, (i + 1) * n / s};
    return w;
}

class Worker {
  public:
    Worker(int device, int curve) : curve_(curve) { check(plonk_create(&ctx_, device, curve)); }
    ~Worker() { plonk_destroy(ctx_); }
    Worker(const Worker&) = delete;
    Worker& operator=(const Worker&) = delete;
    plonk_ctx* ctx() const { return ctx_; }
    int curve() const { return curve_; }

    // init@0 (worker.rs:126-157); bases in PLONK_BASES_XY layout, n_bases points
    void init(const uint64_t* bases_xy, size_t n_bases, size_t domain_size, size_t quot_domain_size) {
        check(plonk_init(ctx_, bases_xy, n_bases, PLONK_BASES_XY, domain_size, quot_domain_size));
    }
    // varMsm@1 (worker.rs:159-185) -> raw Jacobian X||Y||Z
    std::vector<uint64_t> var_msm(const plonk_msm_workload& wl, const uint64_t* scalars, size_t n_scalars) {
        std::vector<uint64_t> out(3 * fq_limbs64(curve_));
        check(plonk_var_msm(ctx_, &wl, scalars, n_scalars, out.data()));
        return out;
    }
    void fft_init(uint64_t id, const std::vector<plonk_fft_workload>& wl, size_t me, bool is_quot, bool is_inv, bool is_coset) {
        check(plonk_fft_init(ctx_, id, wl.data(), wl.size(), me, is_quot, is_inv, is_coset));
    }
    void fft1(uint64_t id, uint64_t i, const uint64_t* row, size_t len) { check(plonk_fft1(ctx_, id, i, row, len)); }
    void fft2_prepare(uint64_t id, plonk_exchange_fn exchange, void* user) { check(plonk_fft2_prepare(ctx_, id, exchange, user)); }
    void fft2(uint64_t id, uint64_t* out_cols) { check(plonk_fft2(ctx_, id, out_cols)); }
    void sync() { check(plonk_sync(ctx_)); }

  private:
    plonk_ctx* ctx_ = nullptr;
    int curve_;
};

class Dispatcher {
  public:
    Dispatcher(size_t num_slaves, int device, int curve, uint64_t seed = 0xD15EA5E) : curve_(curve), rng_(seed) {
        for (size_t i = 0; i < num_slaves; i++) workers_.emplace_back(new Worker(device, curve));
    }
    ~Dispatcher() { for (Worker* w : workers_) delete w; }
    size_t num_slaves() const { return workers_.size(); }
    Worker& worker(size_t i) { return *workers_[i]; }

    // dispatcher.rs:213-216: the full SRS goes to every worker
    void init(const uint64_t* bases_xy, size_t n_bases, size_t domain_size, size_t quot_domain_size) {
        for (Worker* w : workers_) w->init(bases_xy, n_bases, domain_size, quot_domain_size);
        n_bases_ = n_bases; domain_size_ = domain_size; quot_domain_size_ = quot_domain_size;
    }

    // dispatcher.rs:218-238: contiguous shards, reduce(|a, b| a + b).  scalars: canonical, n of them
    std::vector<uint64_t> msm(const uint64_t* scalars, size_t n) {
        std::vector<uint64_t> acc;
        for (size_t i = 0; i < workers_.size(); i++) {
            const plonk_msm_workload wl = make_msm_workloads(n, workers_.size())[i];
            std::vector<uint64_t> part = workers_[i]->var_msm(wl, scalars + 4 * wl.start, wl.end - wl.start);
            if (acc.empty()) acc = part;
            else { std::vector<uint64_t> s(part.size()); check(plonk_g1_add(curve_, acc.data(), part.data(), s.data())); acc = s; }
        }
        return acc;
    }

    // dispatcher2.rs:835-893 -> affine x||y and the infinity flag; coefficients in Montgomery form
    bool commit_polynomial(const uint64_t* coeffs_mont, size_t n_coeffs, std::vector<uint64_t>* xy) {
        const size_t n = n_coeffs < n_bases_ ? n_coeffs : n_bases_;
        std::vector<uint64_t> canon(4 * (n ? n : 1));
        if (n) check(plonk_debug_field_op(workers_[0]->ctx(), 0, 4, coeffs_mont, nullptr, canon.data(), n));      // into_repr
        std::vector<uint64_t> jac = msm_over(canon.data(), n);
        xy->assign(2 * fq_limbs64(curve_), 0);
        int inf = 0;
        check(plonk_g1_to_affine(curve_, jac.data(), xy->data(), &inf));
        return inf != 0;
    }

    // The commitments of ONE prover round — independent polynomials against the same key (dispatcher2.rs:313-321 five wires, :519-531
    // five quotient parts, :690-697 two openings) — where the reference loops commit_polynomial: every worker takes its key range
    // [lo, hi) (dispatcher2.rs:875-878) of ALL polynomials through plonk_commit_many_dev (one Pippenger problem per worker), the
    // partial points are added per polynomial (:887-890).  polys[k]: lens[k] Montgomery coefficients on the host.
    // Returns the infinity flags; xys[k] = affine x||y.
    std::vector<bool> commit_round(const std::vector<const uint64_t*>& polys, const std::vector<size_t>& lens, std::vector<std::vector<uint64_t>>* xys) {
        const size_t K = polys.size(), S = workers_.size(), J = 3 * fq_limbs64(curve_);
        std::vector<std::vector<uint64_t>> acc(K);
        const std::vector<plonk_msm_workload> wl = make_msm_workloads(n_bases_, S);
        for (size_t s = 0; s < S; s++) {
            plonk_ctx* ctx = workers_[s]->ctx();
            const size_t lo = wl[s].start, hi = wl[s].end;
            std::vector<void*> d(K, nullptr);
            std::vector<const void*> ptrs(K);
            std::vector<size_t> cnt(K);
            for (size_t k = 0; k < K; k++) {
                const size_t len = lens[k] < n_bases_ ? lens[k] : n_bases_;
                cnt[k] = len > lo ? (len < hi ? len : hi) - lo : 0;
                check(plonk_dev_alloc(ctx, (cnt[k] ? cnt[k] : 1) * 32, &d[k]));
                if (cnt[k]) check(plonk_memcpy_h2d(ctx, d[k], polys[k] + 4 * lo, cnt[k] * 32));
                ptrs[k] = d[k];
            }
            std::vector<uint64_t> part(K * J);
            const int rc = plonk_commit_many_dev(ctx, K, ptrs.data(), cnt.data(), lo, part.data());
            for (void* q : d) plonk_dev_free(ctx, q);
            check(rc);
            for (size_t k = 0; k < K; k++) {
                std::vector<uint64_t> pk(part.begin() + k * J, part.begin() + (k + 1) * J);
                if (acc[k].empty()) acc[k] = pk;
                else { std::vector<uint64_t> sum(J); check(plonk_g1_add(curve_, acc[k].data(), pk.data(), sum.data())); acc[k] = sum; }
            }
        }
        std::vector<bool> infs(K);
        xys->assign(K, std::vector<uint64_t>(2 * fq_limbs64(curve_), 0));
        for (size_t k = 0; k < K; k++) {
            int inf = 0;
            check(plonk_g1_to_affine(curve_, acc[k].data(), (*xys)[k].data(), &inf));
            infs[k] = inf != 0;
        }
        return infs;
    }

    // Prover::fft, dispatcher2.rs:732-787.  coeffs: len Fr (zero-padded to the domain); result: N Fr, natural order
    std::vector<uint64_t> fft(const uint64_t* coeffs, size_t len, bool is_quot, bool is_inv, bool is_coset) {
        const size_t N = is_quot ? quot_domain_size_ : domain_size_, S = workers_.size();
        size_t r, c;
        split_rc(N, &r, &c);
        const uint64_t id = rng_();                                     // :743 (thread_rng in the reference)
        const std::vector<plonk_fft_workload> wl = make_fft_workloads(N, S);
        std::vector<uint64_t> v(4 * N, 0);                              // coeffs.resize(domain.size())
        std::memcpy(v.data(), coeffs, (len < N ? len : N) * 32);
        for (size_t s = 0; s < S; s++) workers_[s]->fft_init(id, wl, s, is_quot, is_inv, is_coset);
        std::vector<uint64_t> row(4 * c);
        for (size_t s = 0; s < S; s++)                                  // :754-766: t[b][a] = coeffs[a*r + b], one fft1 per row
            for (uint64_t b = wl[s].row_start; b < wl[s].row_end; b++) {
                for (size_t a = 0; a < c; a++) std::memcpy(&row[4 * a], &v[4 * (a * r + b)], 32);
                workers_[s]->fft1(id, b - wl[s].row_start, row.data(), c);
            }
        fft2_prepare_all(id);                                           // :767-772
        std::vector<uint64_t> out(4 * N);
        for (size_t s = 0; s < S; s++) {                                // :774-786: columns back, out[j*c + i] = u[i][j]
            const size_t ncols = wl[s].col_end - wl[s].col_start;
            std::vector<uint64_t> cols(4 * ncols * r);
            workers_[s]->fft2(id, cols.data());
            for (size_t i = 0; i < ncols; i++)
                for (size_t j = 0; j < r; j++) std::memcpy(&out[4 * (j * c + wl[s].col_start + i)], &cols[4 * (i * r + j)], 32);
        }
        return out;
    }

  private:
    struct Recorded { const void* send; void* recv; size_t bytes; };
    static int record(void* user, const void* send, void* recv, size_t bytes_per_peer, int, void*) {
        Recorded* r = static_cast<Recorded*>(user);
        r->send = send; r->recv = recv; r->bytes = bytes_per_peer;
        return 0;
    }
    // every worker runs its row pass and reports its send / receive buffers; block (s -> d) is then copied device to device —
    // the in-process form of the peers' fftExchange (worker.rs:412-438).  One process per GPU uses plonk_comm_init instead.
    void fft2_prepare_all(uint64_t id) {
        const size_t S = workers_.size();
        if (S == 1) { workers_[0]->fft2_prepare(id, nullptr, nullptr); return; }
        std::vector<Recorded> rec(S);
        for (size_t s = 0; s < S; s++) workers_[s]->fft2_prepare(id, &Dispatcher::record, &rec[s]);
        for (Worker* w : workers_) w->sync();
        for (size_t d = 0; d < S; d++)
            for (size_t s = 0; s < S; s++)
                check(plonk_memcpy_d2d(workers_[d]->ctx(), (char*)rec[d].recv + s * rec[s].bytes, (const char*)rec[s].send + d * rec[s].bytes, rec[s].bytes));
        for (Worker* w : workers_) w->sync();
    }
    std::vector<uint64_t> msm_over(const uint64_t* canon, size_t n) {
        if (n == 0) { plonk_msm_workload z{0, 0}; return workers_[0]->var_msm(z, canon, 0); }
        return msm(canon, n);
    }

    std::vector<Worker*> workers_;
    int curve_;
    size_t n_bases_ = 0, domain_size_ = 0, quot_domain_size_ = 0;
    std::mt19937_64 rng_;
};

}  // namespace plonk
