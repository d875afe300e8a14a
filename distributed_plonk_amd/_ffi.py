"""ctypes binding of libplonk_hip.so (include/plonk_hip.h).

The HIP library is the only compute engine of this package: there is no CPU fallback.  If the
shared object is missing, importing this module fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PLONK_HIP_LIB") or os.path.join(_HERE, "lib", "libplonk_hip.so")      # PLONK_HIP_LIB: an alternative build (A/B experiments)

PLONK_BN254, PLONK_BLS12_381 = 0, 1
PLONK_BASES_XY, PLONK_BASES_ARK = 0, 1
CURVES = {"bn254": PLONK_BN254, "bls12_381": PLONK_BLS12_381}
FQ_LIMBS64 = {PLONK_BN254: 4, PLONK_BLS12_381: 6}
ERR_NAMES = {0: "PLONK_OK", -1: "PLONK_ERR_ARG", -2: "PLONK_ERR_DOMAIN", -3: "PLONK_ERR_HIP",
             -4: "PLONK_ERR_STATE", -5: "PLONK_ERR_EXCHANGE"}


class PlonkError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


class FftWorkload(C.Structure):          # reference src/utils.rs:3-8
    _fields_ = [("row_start", C.c_uint64), ("row_end", C.c_uint64), ("col_start", C.c_uint64), ("col_end", C.c_uint64)]

    def num_rows(self) -> int:           # utils.rs:12-14
        return self.row_end - self.row_start

    def num_cols(self) -> int:           # utils.rs:16-18
        return self.col_end - self.col_start

    def __repr__(self):
        return f"FftWorkload(rows=[{self.row_start},{self.row_end}), cols=[{self.col_start},{self.col_end}))"


class MsmWorkload(C.Structure):          # reference src/utils.rs:21-25
    _fields_ = [("start", C.c_uint64), ("end", C.c_uint64)]

    def __repr__(self):
        return f"MsmWorkload([{self.start},{self.end}))"


class QuotientInputs(C.Structure):   # include/plonk_hip.h plonk_quotient_inputs (device pointers)
    _fields_ = [("selectors", C.c_void_p * 13), ("sigmas", C.c_void_p * 5), ("wires", C.c_void_p * 5),
                ("perm", C.c_void_p), ("pub_input", C.c_void_p)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)

# name -> (restype, argtypes); every symbol include/plonk_hip.h declares
PLONK_COMM_ID_BYTES = 128

SIGNATURES = {
    "plonk_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    "plonk_destroy": (None, [C.c_void_p]),
    "plonk_last_error": (C.c_char_p, []),
    "plonk_stream": (C.c_void_p, [C.c_void_p]),
    "plonk_sync": (C.c_int, [C.c_void_p]),
    "plonk_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t]),
    "plonk_var_msm": (C.c_int, [C.c_void_p, C.POINTER(MsmWorkload), C.c_void_p, C.c_size_t, C.c_void_p]),
    "plonk_fft_init": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(FftWorkload), C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int]),
    "plonk_fft1": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t]),
    "plonk_fft2_prepare": (C.c_int, [C.c_void_p, C.c_uint64, EXCHANGE_FN, C.c_void_p]),
    "plonk_fft2": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    "plonk_round1": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "plonk_get_wire": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "plonk_ntt": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int]),
    "plonk_commit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "plonk_g1_add": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "plonk_g1_to_affine": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "plonk_keccak_f1600": (C.c_int, [C.c_void_p]),
    "plonk_transpose": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]),
    "plonk_ntt_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int]),
    "plonk_msm_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]),
    "plonk_commit_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "plonk_commit_range_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]),
    "plonk_commit_many_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "plonk_fft1_dev": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    "plonk_fft1_dev_compact": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t]),
    "plonk_fft2_dev": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]),
    "plonk_transpose_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]),
    "plonk_trim": (C.c_int, [C.c_void_p]),
    "plonk_dev_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "plonk_dev_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "plonk_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "plonk_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "plonk_memcpy_d2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "plonk_synth_fr": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t]),
    "plonk_synth_bases": (C.c_int, [C.c_void_p, C.c_uint64, C.c_size_t, C.c_size_t, C.c_void_p]),
    "plonk_synth_srs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "plonk_synth_circuit": (C.c_int, [C.c_void_p, C.c_uint64, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
    "plonk_init_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t]),
    "plonk_debug_field_op": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "plonk_quotient_evals_dev": (C.c_int, [C.c_void_p, C.POINTER(QuotientInputs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "plonk_perm_product_dev": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p * 5), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "plonk_perm_product_range_dev": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p * 5), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                               C.c_size_t, C.c_void_p]),
    "plonk_class_interleave_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]),
    "plonk_memcpy_d2d_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "plonk_exchange_standin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "plonk_poly_eval_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "plonk_poly_lincomb_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "plonk_poly_div_linear_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "plonk_poly_degree_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int64)]),
    "plonk_memset_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t]),
    "plonk_coset_eval_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]),
    "plonk_coset_interp_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]),
    "plonk_blind_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "plonk_quotient_evals_class_dev": (C.c_int, [C.c_void_p, C.POINTER(QuotientInputs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                                 C.c_uint32, C.c_void_p]),
    "plonk_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "plonk_last_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "plonk_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "plonk_profile_get": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "plonk_profile_reset": (C.c_int, [C.c_void_p]),
    "plonk_comm_unique_id": (C.c_int, [C.c_void_p]),
    "plonk_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "plonk_comm_destroy": (C.c_int, [C.c_void_p]),
    "plonk_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "plonk_exchange_rccl": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "plonk_comm_alltoall_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "plonk_comm_allgather_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "plonk_comm_allgather_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
}

_lib = None


def lib():
    """Load libplonk_hip.so.  torch (if used in this process) must be imported first so both share
    one HIP runtime; we import it here when available to fix the order."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension is the only compute path of distributed_plonk_amd "
                "(no CPU fallback). Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `python -m distributed_plonk_amd.build`.")
        try:
            import torch  # noqa: F401  (loads torch's libamdhip64 first when torch is installed)
        except Exception:  # pragma: no cover - torch is optional for the C ABI itself
            pass
        L = C.CDLL(LIB_PATH)
        if hasattr(L, "plonk_hostemu_marker") and os.environ.get("PLONK_ALLOW_HOSTEMU") != "1":
            # tests/hostemu builds the kernel sources for the CPU so the test suite can execute them without a GPU; that library is
            # test infrastructure and never a compute path of this package: refuse it unless the test harness itself opted in
            raise ImportError(f"{LIB_PATH} is the host EMULATION of libplonk_hip.so (tests/hostemu): distributed_plonk_amd has no CPU "
                              "path. Unset PLONK_HIP_LIB, or set PLONK_ALLOW_HOSTEMU=1 if you are the test harness.")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError here = header/library mismatch: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int):
    if rc != 0:
        raise PlonkError(rc, lib().plonk_last_error().decode(errors="replace"))
