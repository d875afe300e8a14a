//! plonk_hip.rs — Rust binding of `include/plonk_hip.h` (libplonk_hip.so), to be dropped into the reference as
//! `src/plonk_hip.rs` (`mod plonk_hip;` in `src/worker.rs`).  `build.rs` gains
//!     println!("cargo:rustc-link-search=native=<repo>/distributed_plonk_amd/lib");
//!     println!("cargo:rustc-link-lib=dylib=plonk_hip");
//!
//! This image has no Rust toolchain, so this file is written against the header, not compiled; `tests/test_abi_and_host_logic.py`
//! checks that every function the header declares is bound here with the same number of arguments.  The thin safe layer at the
//! bottom is what the handlers of `impl plonk_slave::Server for PlonkImpl` (worker.rs:125-439) call: each replaces the arkworks
//! call cited next to it.
#![allow(non_camel_case_types, dead_code)]

use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct plonk_ctx {
    _private: [u8; 0],
}

pub const PLONK_BN254: c_int = 0;
pub const PLONK_BLS12_381: c_int = 1;
pub const PLONK_BASES_XY: c_int = 0;
pub const PLONK_BASES_ARK: c_int = 1;
pub const PLONK_OK: c_int = 0;
pub const PLONK_ERR_ARG: c_int = -1;
pub const PLONK_ERR_DOMAIN: c_int = -2;
pub const PLONK_ERR_HIP: c_int = -3;
pub const PLONK_ERR_STATE: c_int = -4;
pub const PLONK_ERR_EXCHANGE: c_int = -5;
pub const PLONK_COMM_ID_BYTES: usize = 128;

/// utils.rs:3-8 / hello_world.capnp:8-13
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct plonk_fft_workload {
    pub row_start: u64,
    pub row_end: u64,
    pub col_start: u64,
    pub col_end: u64,
}

/// utils.rs:21-25 / hello_world.capnp:3-6
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct plonk_msm_workload {
    pub start: u64,
    pub end: u64,
}

/// dispatcher2.rs:443-456 (selector order), :382-432 (the 25 coset-FFT outputs)
#[repr(C)]
pub struct plonk_quotient_inputs {
    pub selectors: [*const c_void; 13],
    pub sigmas: [*const c_void; 5],
    pub wires: [*const c_void; 5],
    pub perm: *const c_void,
    pub pub_input: *const c_void,
}

pub type plonk_exchange_fn = Option<
    unsafe extern "C" fn(user: *mut c_void, send: *const c_void, recv: *mut c_void, bytes_per_peer: usize, n_ranks: c_int, stream: *mut c_void) -> c_int,
>;

#[link(name = "plonk_hip")]
extern "C" {
    // ---- in-library RCCL transport (replaces the TCP + Cap'n Proto peer links of worker.rs:280-345, 412-438)
    pub fn plonk_comm_unique_id(out_id: *mut c_void) -> c_int;
    pub fn plonk_comm_init(ctx: *mut plonk_ctx, id: *const c_void, rank: c_int, world: c_int) -> c_int;
    pub fn plonk_comm_destroy(ctx: *mut plonk_ctx) -> c_int;
    pub fn plonk_comm_info(ctx: *mut plonk_ctx, rank: *mut c_int, world: *mut c_int, rccl_version: *mut c_int) -> c_int;
    pub fn plonk_exchange_rccl(user: *mut c_void, send: *const c_void, recv: *mut c_void, bytes_per_peer: usize, n_ranks: c_int, stream: *mut c_void) -> c_int;
    pub fn plonk_exchange_standin(user: *mut c_void, send: *const c_void, recv: *mut c_void, bytes_per_peer: usize, n_ranks: c_int, stream: *mut c_void) -> c_int;
    pub fn plonk_comm_alltoall_dev(ctx: *mut plonk_ctx, d_send: *const c_void, d_recv: *mut c_void, bytes_per_peer: usize) -> c_int;
    pub fn plonk_comm_allgather_dev(ctx: *mut plonk_ctx, d_send: *const c_void, d_recv: *mut c_void, bytes: usize) -> c_int;
    pub fn plonk_comm_allgather_host(ctx: *mut plonk_ctx, input: *const c_void, bytes: usize, out: *mut c_void) -> c_int;

    // ---- lifetime (State::new, worker.rs:455-472)
    pub fn plonk_create(out: *mut *mut plonk_ctx, device: c_int, curve: c_int) -> c_int;
    pub fn plonk_destroy(ctx: *mut plonk_ctx);
    pub fn plonk_last_error() -> *const c_char;
    pub fn plonk_stream(ctx: *mut plonk_ctx) -> *mut c_void;
    pub fn plonk_sync(ctx: *mut plonk_ctx) -> c_int;

    // ---- PlonkSlave @0..@6 + PlonkPeer @0
    pub fn plonk_init(ctx: *mut plonk_ctx, bases: *const c_void, n_bases: usize, base_layout: c_int, domain_size: usize, quot_domain_size: usize) -> c_int; // worker.rs:126-157
    pub fn plonk_var_msm(ctx: *mut plonk_ctx, workload: *const plonk_msm_workload, scalars: *const u64, n_scalars: usize, out_jacobian: *mut u64) -> c_int; // worker.rs:159-185
    pub fn plonk_fft_init(ctx: *mut plonk_ctx, id: u64, workloads: *const plonk_fft_workload, n_workloads: usize, me: usize, is_quot: c_int, is_inv: c_int, is_coset: c_int) -> c_int; // worker.rs:187-233
    pub fn plonk_fft1(ctx: *mut plonk_ctx, id: u64, i: u64, v: *const u64, len: usize) -> c_int; // worker.rs:235-278
    pub fn plonk_fft2_prepare(ctx: *mut plonk_ctx, id: u64, exchange: plonk_exchange_fn, user: *mut c_void) -> c_int; // worker.rs:280-345 + 412-438
    pub fn plonk_fft2(ctx: *mut plonk_ctx, id: u64, out_cols: *mut u64) -> c_int; // worker.rs:347-381
    pub fn plonk_round1(ctx: *mut plonk_ctx, evals: *const u64, n: usize, blinders: *const u64, out_commit_jacobian: *mut u64) -> c_int; // worker.rs:383-408
    pub fn plonk_get_wire(ctx: *mut plonk_ctx, out_coeffs: *mut u64, n_coeffs: usize) -> c_int; // state.wire, worker.rs:58

    // ---- the third-party operator calls (ark-poly / ark-ec), host buffers
    pub fn plonk_ntt(ctx: *mut plonk_ctx, v: *mut u64, n: usize, is_inv: c_int, is_coset: c_int) -> c_int; // dispatcher.rs:594,632,667; dispatcher2.rs:507
    pub fn plonk_commit(ctx: *mut plonk_ctx, coeffs_mont: *const u64, n_coeffs: usize, out_jacobian: *mut u64) -> c_int; // worker.rs:117-123
    pub fn plonk_g1_add(curve: c_int, a_jac: *const u64, b_jac: *const u64, out_jac: *mut u64) -> c_int; // dispatcher.rs:236-238
    /// Keccak-f[1600] in place on a 200-byte state (merlin's permutation; a Rust host keeps using the merlin crate).
    pub fn plonk_keccak_f1600(state200: *mut u8) -> c_int;
    pub fn plonk_g1_to_affine(curve: c_int, jac: *const u64, out_xy: *mut u64, is_infinity: *mut c_int) -> c_int; // dispatcher2.rs:892
    pub fn plonk_transpose(ctx: *mut plonk_ctx, v: *mut u64, rows: usize, cols: usize) -> c_int; // transpose.rs:413

    // ---- device-resident variants
    pub fn plonk_ntt_dev(ctx: *mut plonk_ctx, d_in: *mut c_void, d_out: *mut c_void, n: usize, is_inv: c_int, is_coset: c_int) -> c_int;
    pub fn plonk_msm_dev(ctx: *mut plonk_ctx, start: usize, end: usize, d_scalars: *const c_void, out_jacobian: *mut u64) -> c_int;
    pub fn plonk_commit_dev(ctx: *mut plonk_ctx, d_coeffs_mont: *const c_void, n_coeffs: usize, out_jacobian: *mut u64) -> c_int;
    pub fn plonk_commit_range_dev(ctx: *mut plonk_ctx, d_coeffs_mont: *const c_void, start: usize, count: usize, out_jacobian: *mut u64) -> c_int; // dispatcher2.rs:870-890
    pub fn plonk_commit_many_dev(ctx: *mut plonk_ctx, k: usize, d_coeffs_mont: *const *const c_void, n_coeffs: *const usize, start: usize, out_jacobians: *mut u64) -> c_int; // the commitments of a round in one launch set
    pub fn plonk_fft1_dev(ctx: *mut plonk_ctx, id: u64, d_rows: *mut c_void) -> c_int;
    pub fn plonk_fft1_dev_compact(ctx: *mut plonk_ctx, id: u64, d_rows: *const c_void, row_len: usize) -> c_int; // dispatcher2.rs:746,754: zero-padded rows
    pub fn plonk_fft2_dev(ctx: *mut plonk_ctx, id: u64, d_out: *mut c_void, layout: c_int) -> c_int;
    pub fn plonk_transpose_dev(ctx: *mut plonk_ctx, d_in: *const c_void, d_out: *mut c_void, rows: usize, cols: usize) -> c_int;

    // ---- the dispatcher's own O(n) loops (dispatcher2.rs:329-344, 435-504, 545-688)
    pub fn plonk_quotient_evals_dev(ctx: *mut plonk_ctx, input: *const plonk_quotient_inputs, alpha: *const u64, beta: *const u64, gamma: *const u64, k: *const u64, d_out: *mut c_void) -> c_int;
    pub fn plonk_quotient_evals_class_dev(ctx: *mut plonk_ctx, input: *const plonk_quotient_inputs, alpha: *const u64, beta: *const u64, gamma: *const u64, k: *const u64, class_stride: u32, class_offset: u32, d_out: *mut c_void) -> c_int;
    pub fn plonk_perm_product_dev(ctx: *mut plonk_ctx, d_wires: *const *const c_void, d_id_perm: *const c_void, d_perm_idx: *const c_void, beta: *const u64, gamma: *const u64, n: usize, d_out: *mut c_void) -> c_int;
    pub fn plonk_perm_product_range_dev(ctx: *mut plonk_ctx, d_wires: *const *const c_void, d_id_perm: *const c_void, d_perm_idx: *const c_void, beta: *const u64, gamma: *const u64, n: usize, first: usize, count: usize, d_out: *mut c_void) -> c_int;
    pub fn plonk_class_interleave_dev(ctx: *mut plonk_ctx, d_in: *const c_void, classes: usize, size: usize, in_stride: usize, reverse: c_int, scale: *const u64, d_out: *mut c_void) -> c_int;
    pub fn plonk_poly_eval_dev(ctx: *mut plonk_ctx, d_poly: *const c_void, len: usize, point: *const u64, out: *mut u64) -> c_int;
    pub fn plonk_poly_lincomb_dev(ctx: *mut plonk_ctx, k: usize, d_polys: *const *const c_void, lens: *const usize, coeffs: *const u64, d_out: *mut c_void, out_len: usize) -> c_int;
    pub fn plonk_poly_div_linear_dev(ctx: *mut plonk_ctx, d_poly: *const c_void, len: usize, point: *const u64, d_out: *mut c_void) -> c_int;
    pub fn plonk_poly_degree_dev(ctx: *mut plonk_ctx, d_poly: *const c_void, len: usize, degree: *mut i64) -> c_int;
    pub fn plonk_blind_dev(ctx: *mut plonk_ctx, d_poly: *mut c_void, n: usize, blinders: *const u64, k: usize) -> c_int;
    pub fn plonk_coset_eval_dev(ctx: *mut plonk_ctx, d_poly: *const c_void, len: usize, size: usize, shift: *const u64, d_out: *mut c_void) -> c_int;
    pub fn plonk_coset_interp_dev(ctx: *mut plonk_ctx, d_evals: *mut c_void, size: usize, shift: *const u64, scale: *const u64, i0: usize, count: usize, d_out: *mut c_void) -> c_int;

    // ---- device memory, synthetic inputs, knobs, timing
    /// Release every cache the context can rebuild on demand (finished tasks' exchange buffers, NTT planes, MSM workspace, scratch).
    pub fn plonk_trim(ctx: *mut plonk_ctx) -> c_int;
    pub fn plonk_dev_alloc(ctx: *mut plonk_ctx, bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn plonk_dev_free(ctx: *mut plonk_ctx, p: *mut c_void) -> c_int;
    pub fn plonk_memcpy_h2d(ctx: *mut plonk_ctx, d_dst: *mut c_void, h_src: *const c_void, bytes: usize) -> c_int;
    pub fn plonk_memcpy_d2h(ctx: *mut plonk_ctx, h_dst: *mut c_void, d_src: *const c_void, bytes: usize) -> c_int;
    pub fn plonk_memcpy_d2d(ctx: *mut plonk_ctx, d_dst: *mut c_void, d_src: *const c_void, bytes: usize) -> c_int;
    pub fn plonk_memcpy_d2d_async(ctx: *mut plonk_ctx, d_dst: *mut c_void, d_src: *const c_void, bytes: usize) -> c_int;
    pub fn plonk_memset_dev(ctx: *mut plonk_ctx, d_dst: *mut c_void, byte: c_int, bytes: usize) -> c_int;
    pub fn plonk_synth_fr(ctx: *mut plonk_ctx, seed: u64, d_out: *mut c_void, n: usize) -> c_int;
    pub fn plonk_synth_bases(ctx: *mut plonk_ctx, seed: u64, unique: usize, n: usize, d_out: *mut c_void) -> c_int;
    pub fn plonk_synth_srs(ctx: *mut plonk_ctx, tau: *const u64, n: usize, d_out: *mut c_void) -> c_int;
    pub fn plonk_synth_circuit(ctx: *mut plonk_ctx, seed: u64, n: usize, num_inputs: usize, k: *const u64, d_wires: *mut c_void, d_selector_evals: *mut c_void,
                               d_sigma_evals: *mut c_void, d_id_perm: *mut c_void, d_perm_idx: *mut c_void, d_pub_input: *mut c_void) -> c_int;
    pub fn plonk_init_dev(ctx: *mut plonk_ctx, d_bases_xy: *const c_void, n_bases: usize, domain_size: usize, quot_domain_size: usize) -> c_int;
    pub fn plonk_debug_field_op(ctx: *mut plonk_ctx, field: c_int, op: c_int, a: *const u64, b: *const u64, out: *mut u64, n: usize) -> c_int;
    pub fn plonk_set_option(ctx: *mut plonk_ctx, key: *const c_char, value: i64) -> c_int;
    pub fn plonk_last_kernel_ms(ctx: *mut plonk_ctx, out_ms: *mut f64) -> c_int;
    pub fn plonk_profile_enable(ctx: *mut plonk_ctx, on: c_int) -> c_int;
    pub fn plonk_profile_reset(ctx: *mut plonk_ctx) -> c_int;
    pub fn plonk_profile_get(ctx: *mut plonk_ctx, name: *const c_char, total_ms: *mut f64, launches: *mut u64) -> c_int;
}

// ------------------------------------------------------------------------------------------------ safe layer for worker.rs
#[derive(Debug)]
pub struct PlonkError {
    pub code: c_int,
    pub message: String,
}

fn check(rc: c_int) -> Result<(), PlonkError> {
    if rc == PLONK_OK {
        return Ok(());
    }
    let message = unsafe { CStr::from_ptr(plonk_last_error()) }.to_string_lossy().into_owned();
    Err(PlonkError { code: rc, message })
}

/// Replaces `State` (worker.rs:42-59): SRS, domains, FFT tasks and `wire` live on the GPU inside the context.
pub struct GpuState {
    ctx: *mut plonk_ctx,
}

impl GpuState {
    /// `State::new` (worker.rs:455-472).  One per GPU; like the reference worker, not thread-safe.
    pub fn new(device: i32, curve: c_int) -> Result<Self, PlonkError> {
        let mut ctx = std::ptr::null_mut();
        check(unsafe { plonk_create(&mut ctx, device, curve) })?;
        Ok(GpuState { ctx })
    }

    /// Join the RCCL communicator of the worker set (`config/network.json` gives rank and world; rank 0 creates the id with
    /// `plonk_comm_unique_id` and hands it out over the existing dispatcher connection).  Collective.
    pub fn join(&mut self, id: &[u8; PLONK_COMM_ID_BYTES], rank: i32, world: i32) -> Result<(), PlonkError> {
        check(unsafe { plonk_comm_init(self.ctx, id.as_ptr() as *const c_void, rank, world) })
    }

    /// `init` (worker.rs:126-157): `bases` is the concatenated `serialize(&[G1Affine])` payload exactly as it arrives.
    pub fn init(&mut self, bases: &[u8], affine_size: usize, domain_size: usize, quot_domain_size: usize) -> Result<(), PlonkError> {
        check(unsafe { plonk_init(self.ctx, bases.as_ptr() as *const c_void, bases.len() / affine_size, PLONK_BASES_ARK, domain_size, quot_domain_size) })
    }

    /// `var_msm` (worker.rs:159-185): `VariableBaseMSM::multi_scalar_mul(&bases[start..end], &scalars)`; the reply is the raw
    /// `G1Projective` (X || Y || Z, Montgomery) the dispatcher adds up (dispatcher.rs:236-238).
    pub fn var_msm(&mut self, start: u64, end: u64, scalars: &[u64], out_jacobian: &mut [u64]) -> Result<(), PlonkError> {
        let wl = plonk_msm_workload { start, end };
        check(unsafe { plonk_var_msm(self.ctx, &wl, scalars.as_ptr(), scalars.len() / 4, out_jacobian.as_mut_ptr()) })
    }

    /// `fft_init` (worker.rs:187-233)
    pub fn fft_init(&mut self, id: u64, workloads: &[plonk_fft_workload], me: usize, is_quot: bool, is_inv: bool, is_coset: bool) -> Result<(), PlonkError> {
        check(unsafe { plonk_fft_init(self.ctx, id, workloads.as_ptr(), workloads.len(), me, is_quot as c_int, is_inv as c_int, is_coset as c_int) })
    }

    /// `fft1` (worker.rs:235-278): one decimated row; the row pass itself (fft1_helper, :66-94) runs for all rows in `fft2_prepare`.
    pub fn fft1(&mut self, id: u64, i: u64, row: &[u64]) -> Result<(), PlonkError> {
        check(unsafe { plonk_fft1(self.ctx, id, i, row.as_ptr(), row.len() / 4) })
    }

    /// `fft2_prepare` (worker.rs:280-345) + the peers' `fft_exchange` (:412-438): row pass, then ONE RCCL all-to-all inside the
    /// library (grouped ncclSend/ncclRecv on the context's stream).  Every rank calls it for the same id.
    pub fn fft2_prepare(&mut self, id: u64) -> Result<(), PlonkError> {
        check(unsafe { plonk_fft2_prepare(self.ctx, id, None, std::ptr::null_mut()) })
    }

    /// `fft2` (worker.rs:347-381): column pass; `out_cols` = num_cols blobs of r elements (the reply of :366-376).
    pub fn fft2(&mut self, id: u64, out_cols: &mut [u64]) -> Result<(), PlonkError> {
        check(unsafe { plonk_fft2(self.ctx, id, out_cols.as_mut_ptr()) })
    }

    /// `round1` (worker.rs:383-408): the two blinders are drawn by the caller from `state.rng` as before.
    pub fn round1(&mut self, evals: &[u64], blinders: &[u64; 8], out_commit_jacobian: &mut [u64]) -> Result<(), PlonkError> {
        check(unsafe { plonk_round1(self.ctx, evals.as_ptr(), evals.len() / 4, blinders.as_ptr(), out_commit_jacobian.as_mut_ptr()) })
    }

    pub fn raw(&self) -> *mut plonk_ctx {
        self.ctx
    }
}

impl Drop for GpuState {
    fn drop(&mut self) {
        unsafe { plonk_destroy(self.ctx) }
    }
}
