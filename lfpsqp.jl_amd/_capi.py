"""ctypes binding of liblfpsqp_hip.so (include/lfpsqp_hip.h) -- the same C ABI a
Julia host binds with ccall.  There is NO CPU fallback: if the HIP
library is missing or no GPU is visible, everything here fails loudly."""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "lib", "liblfpsqp_hip.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "lfpsqp_hip.h")

c_i64 = C.c_int64
c_dbl = C.c_double
P = C.c_void_p
PD = C.POINTER(C.c_double)


class DiagOp(C.Structure):  # lfpsqp_diag_op
    _fields_ = [("a0", c_dbl), ("dg", P)]


class LowRankOp(C.Structure):  # lfpsqp_lowrank_op
    _fields_ = [("a0", c_dbl), ("dg", P), ("V", P), ("k", c_i64), ("sigma", P)]


class TridiagOp(C.Structure):  # lfpsqp_tridiag_op
    _fields_ = [("a0", c_dbl), ("dg", P), ("off", P)]


class Basis(C.Structure):  # lfpsqp_basis
    _fields_ = [("Z", P), ("ncols", c_i64), ("Dx", P), ("Dy", P), ("sx", P), ("sy", P), ("A", P), ("W", P), ("S", P), ("SA", P)]


class IneqData(C.Structure):  # lfpsqp_ineq_data
    _fields_ = [("q", P), ("r", P), ("s", P), ("t", P), ("n", c_i64)]


class Elementwise(C.Structure):  # lfpsqp_elementwise
    _fields_ = [("A", P), ("Asp", P), ("kind", P), ("qw", P), ("work", P)]


class Constraints(C.Structure):  # lfpsqp_constraints
    _fields_ = [("Jct", P), ("m_lin", c_i64), ("b", P), ("has_ball", C.c_int), ("R2", c_dbl), ("n_x", c_i64), ("slack_row", c_i64), ("Jsp", P),
                ("ew", C.POINTER(Elementwise))]


CFUN = C.CFUNCTYPE(C.c_int, P, P, PD)
JACFUN = C.CFUNCTYPE(C.c_int, P, P, P, PD)


class PPWork(C.Structure):  # lfpsqp_pp_work
    _fields_ = [(k, P) for k in ("r", "p", "z", "dx", "g", "tmp_m", "tmp_w", "h", "DxS", "DyS", "ones", "zeros", "q", "i11", "i12", "i22")] + \
               [("precondition", C.c_int)]


class PcgPrecond(C.Structure):  # lfpsqp_pcg_precond
    _fields_ = [("K", P), ("i11", P), ("i12", P), ("i22", P), ("q", P)]


class ProjCGWorkC(C.Structure):  # lfpsqp_projcg_work
    _fields_ = [("g", P), ("d", P), ("rp", P), ("Utr", P)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, P, P, c_i64, C.c_int, P)
OPFUN = C.CFUNCTYPE(C.c_int, P, P, P)          # lfpsqp_opfun(user, src, dest)

_SIGS = {
    "lfpsqp_ctx_create": [C.c_int, C.POINTER(P)],
    "lfpsqp_ctx_destroy": [P],
    "lfpsqp_ctx_sync": [P],
    "lfpsqp_ctx_set_tuning": [P, C.c_int, C.c_int],
    "lfpsqp_ctx_set_onepass": [P, C.c_int],
    "lfpsqp_factored_basis_supported": [P, P, P, C.POINTER(C.c_int)],
    "lfpsqp_ctx_set_residual_buffers": [P, C.c_int],
    "lfpsqp_device_name": [P, C.c_char_p, c_i64],
    "lfpsqp_device_uuid": [P, C.c_char_p, c_i64],
    "lfpsqp_timer_begin": [P],
    "lfpsqp_timer_end": [P, PD],
    "lfpsqp_shard_range": [c_i64, C.c_int, C.c_int, C.POINTER(c_i64), C.POINTER(c_i64)],
    "lfpsqp_comm_unique_id": [P, P],
    "lfpsqp_comm_init_rccl": [P, C.c_int, C.c_int, P],
    "lfpsqp_comm_init_callback": [P, C.c_int, C.c_int, ALLREDUCE_FN, P],
    "lfpsqp_comm_p2p_export": [P, P],
    "lfpsqp_comm_init_p2p": [P, C.c_int, C.c_int, P],
    "lfpsqp_comm_p2p_allow_coarse": [P, C.c_int],
    "lfpsqp_comm_p2p_info": [P, C.POINTER(C.c_int), C.POINTER(C.c_ulonglong)],
    "lfpsqp_comm_info": [P, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "lfpsqp_vec_alloc": [P, c_i64, C.POINTER(P)],
    "lfpsqp_vec_free": [P, P],
    "lfpsqp_vec_upload": [P, P, c_i64, P, c_i64],
    "lfpsqp_vec_download": [P, P, c_i64, P, c_i64],
    "lfpsqp_vec_fill": [P, P, c_dbl],
    "lfpsqp_vec_copy": [P, P, P],
    "lfpsqp_vec_copy_range": [P, P, c_i64, P, c_i64, c_i64],
    "lfpsqp_mat_alloc": [P, c_i64, c_i64, C.POINTER(P)],
    "lfpsqp_mat_free": [P, P],
    "lfpsqp_ctx_set_placement": [P, C.c_int],
    "lfpsqp_ctx_set_nr_batch_mode": [P, C.c_int],
    "lfpsqp_mat_alloc_placed": [P, c_i64, c_i64, C.POINTER(P)],
    "lfpsqp_vecs_alloc_placed": [P, P, c_i64, c_i64, C.c_int, C.POINTER(P)],
    "lfpsqp_basis_work_alloc_placed": [P, c_i64, c_i64, c_i64, C.c_int, C.POINTER(P), C.POINTER(P)],
    "lfpsqp_placement_info": [P, C.POINTER(C.c_int), C.POINTER(C.c_int), PD, C.c_int],
    "lfpsqp_placement_probe": [P, P, c_i64, P, P, P, C.c_int, PD],
    "lfpsqp_mat_shape": [P, C.POINTER(c_i64), C.POINTER(c_i64)],
    "lfpsqp_mat_upload": [P, P, c_i64, c_i64, P, c_i64],
    "lfpsqp_mat_download": [P, P, c_i64, c_i64, P, c_i64],
    "lfpsqp_mat_copy": [P, P, P],
    "lfpsqp_mat_rowscaled_view": [P, P, P, C.POINTER(P)],
    "lfpsqp_mat_view": [P, P, P, P, P, C.POINTER(P)],
    "lfpsqp_factorize_hint": [P, P, c_i64],
    "lfpsqp_vec_hash_fill": [P, P, C.c_uint64, c_i64, c_dbl, c_dbl],
    "lfpsqp_mat_hash_fill": [P, P, C.c_uint64, c_i64, c_i64, c_dbl, c_i64, c_i64],
    "lfpsqp_gemv_t": [P, P, c_i64, P, P],
    "lfpsqp_gemv_n": [P, P, c_i64, c_dbl, P, c_dbl, P],
    "lfpsqp_dot": [P, P, P, PD],
    "lfpsqp_dot_head": [P, P, P, c_i64, PD],
    "lfpsqp_nrm2": [P, P, PD],
    "lfpsqp_amax": [P, P, PD],
    "lfpsqp_axpby": [P, c_dbl, P, c_dbl, P],
    "lfpsqp_waxpby": [P, c_dbl, P, c_dbl, P, P],
    "lfpsqp_vmul": [P, P, P, P],
    "lfpsqp_vec_fill_range": [P, P, c_i64, c_i64, c_dbl],
    "lfpsqp_affine_head": [P, c_dbl, P, c_dbl, c_i64, P],
    "lfpsqp_sumsq_shift": [P, P, c_i64, c_dbl, PD],
    "lfpsqp_allreduce": [P, P, c_i64],
    "lfpsqp_ineq_data_build": [P, P, P, P, P, P, P],
    "lfpsqp_generate_initial_y": [P, P, C.POINTER(IneqData)],
    "lfpsqp_calculate_h": [P, P, P, C.POINTER(IneqData), PD],
    "lfpsqp_inequality_gradient": [P, P, C.POINTER(IneqData), P, P, P, P, P],
    "lfpsqp_y_retract": [P, P, P, C.POINTER(IneqData)],
    "lfpsqp_calculate_lambda_y": [P, P, c_i64, P, P, P, P, P],
    "lfpsqp_augmented_diag": [P, P, P, C.POINTER(IneqData), P],
    "lfpsqp_q_gemv_t": [P, C.POINTER(Basis), P, P, P],
    "lfpsqp_q_gemv_n": [P, C.POINTER(Basis), c_dbl, P, P, c_dbl, P],
    "lfpsqp_constraints_eval": [P, C.POINTER(Constraints), P, PD],
    "lfpsqp_constraints_jac": [P, C.POINTER(Constraints), P, P, PD],
    "lfpsqp_constraints_hess_diag": [P, C.POINTER(Constraints), P, PD, P],
    "lfpsqp_spmat_clone": [P, P, C.POINTER(P)],
    "lfpsqp_spmat_rowscale": [P, P, P, P],
    "lfpsqp_retract_nr": [P, C.POINTER(Basis), P, P, c_i64, C.POINTER(Constraints), CFUN, P, C.POINTER(IneqData), P, P, P, c_dbl, c_i64,
                          PD, C.POINTER(C.c_int), C.POINTER(c_i64)],
    "lfpsqp_retract_nr_batch_width": [P, C.POINTER(Basis), C.POINTER(Constraints), C.POINTER(C.c_int)],
    "lfpsqp_retract_nr_batch": [P, C.POINTER(Basis), P, P, c_i64, C.POINTER(Constraints), C.POINTER(IneqData), C.c_int, C.POINTER(P), P,
                                C.POINTER(P), c_dbl, c_i64, PD, C.POINTER(C.c_int), C.POINTER(c_i64)],
    "lfpsqp_pcg_pre": [P, c_dbl, C.POINTER(Basis), C.POINTER(PcgPrecond), P, P, P, P, c_dbl, c_i64, C.POINTER(C.c_int), C.POINTER(c_i64)],
    "lfpsqp_pcg": [P, c_dbl, C.POINTER(Basis), P, P, P, P, P, P, c_dbl, c_i64, C.POINTER(C.c_int), C.POINTER(c_i64)],
    "lfpsqp_retract_pp": [P, C.POINTER(Constraints), CFUN, JACFUN, P, P, c_i64, C.POINTER(IneqData), P, P, P, P, P, P, c_dbl, c_dbl, c_i64,
                          c_i64, C.POINTER(PPWork), PD, C.POINTER(C.c_int), C.POINTER(c_i64), C.POINTER(c_i64)],
    "lfpsqp_separable": [P, C.c_int, C.c_int, P, c_dbl, P, c_dbl, P, c_i64, P, PD],
    "lfpsqp_spmat_create": [P, c_i64, c_i64, c_i64, P, P, P, C.POINTER(P)],
    "lfpsqp_spmat_free": [P, P],
    "lfpsqp_spmat_info": [P, C.POINTER(c_i64), C.POINTER(c_i64), C.POINTER(c_i64), C.POINTER(c_i64)],
    "lfpsqp_spmv_t": [P, P, P, P],
    "lfpsqp_spmv_n": [P, P, c_dbl, P, c_dbl, P],
    "lfpsqp_spmat_to_dense": [P, P, P],
    "lfpsqp_gram": [P, P, c_i64, P, P],
    "lfpsqp_rmul": [P, P, c_i64, P, c_i64, P],
    "lfpsqp_factorize": [P, P, P, P, P, P, P, C.POINTER(c_i64), c_dbl],
    "lfpsqp_factorize_rhs": [P, P, P, P, P, P, P, C.POINTER(c_i64), c_dbl, P, P, P],
    "lfpsqp_gram_rhs": [P, P, c_i64, P, c_i64, C.POINTER(P), P, P],
    "lfpsqp_tangent_step": [P, C.POINTER(Basis), P, P, c_i64, P, P, P, C.POINTER(Constraints), P, P, C.POINTER(IneqData), P, P, P,
                            C.POINTER(ProjCGWorkC), C.c_int, P, P, PD],
    "lfpsqp_ineq_rhs": [P, P, P, P, P],
    "lfpsqp_spmat_gram": [P, P, P, P, P],
    "lfpsqp_factorize_sp": [P, P, P, P, P, P, P, P, C.POINTER(c_i64), c_dbl],
    "lfpsqp_small_svd": [P, c_i64, c_i64, P, P, P, P],
    "lfpsqp_projcg": [P, P, P, C.POINTER(DiagOp), C.POINTER(Basis), P, P, c_dbl, c_i64, c_i64, C.c_int,
                      C.POINTER(ProjCGWorkC), C.POINTER(c_i64), PD],
    "lfpsqp_projcg_lowrank": [P, P, P, C.POINTER(LowRankOp), C.POINTER(Basis), P, P, c_dbl, c_i64, c_i64, C.c_int,
                              C.POINTER(ProjCGWorkC), C.POINTER(c_i64), PD],
    "lfpsqp_projcg_tridiag": [P, P, P, C.POINTER(TridiagOp), P, C.POINTER(Basis), P, P, c_dbl, c_i64, c_i64, C.c_int,
                              C.POINTER(ProjCGWorkC), C.POINTER(c_i64), PD],
    "lfpsqp_tridiag_mul": [P, C.POINTER(TridiagOp), P, P],
    "lfpsqp_projcg_op": [P, P, P, P, P, P, C.POINTER(Basis), P, P, c_dbl, c_i64, c_i64, C.c_int,
                         C.POINTER(ProjCGWorkC), C.POINTER(c_i64), PD],
    "lfpsqp_ctx_stream": [P, C.POINTER(P)],
    "lfpsqp_ctx_set_profiling": [P, C.c_int],
    "lfpsqp_profile_read": [P, PD, C.POINTER(c_i64)],
}


class LfpsqpError(RuntimeError):
    pass


def header_functions(header: str = HEADER):
    """Names of every function include/lfpsqp_hip.h declares."""
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lfpsqp_[a-z0-9_]+)\s*\(", text)) - {"lfpsqp_allreduce_fn", "lfpsqp_cfun", "lfpsqp_jacfun"})


class Library:
    def __init__(self, path: str | None = None):
        path = path or DEFAULT_LIB
        if not os.path.exists(path):
            raise LfpsqpError(
                f"HIP extension not built: {path} is missing (run `python -c 'import __graft_entry__ as g; g.build()'`). "
                "There is no CPU fallback.")
        self.path = path
        self.lib = C.CDLL(path)
        self.lib.lfpsqp_last_error.restype = C.c_char_p
        self.lib.lfpsqp_last_error.argtypes = [P]
        self.lib.lfpsqp_vec_len.restype = c_i64
        self.lib.lfpsqp_vec_len.argtypes = [P]
        self.lib.lfpsqp_half_stride.restype = c_i64
        self.lib.lfpsqp_half_stride.argtypes = [c_i64]
        for name, sig in _SIGS.items():
            fn = getattr(self.lib, name)
            fn.restype = C.c_int
            fn.argtypes = sig

    def __getattr__(self, name):
        return getattr(self.lib, name)


_default: Library | None = None


def load_library(path: str | None = None) -> Library:
    """Load (once) the HIP library.  ``path`` is for the test-suite's emulator build only."""
    global _default
    if path is not None:
        return Library(path)
    if _default is None:
        _default = Library()
    return _default
