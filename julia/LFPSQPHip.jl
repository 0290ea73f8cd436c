# LFPSQPHip.jl -- what a maintainer of ksil/LFPSQP.jl adds to run the package's inner loop (projcg!, retract!, the tangent
# setup, the bound operators) AND the outer `optimize` driver on an MI355X through liblfpsqp_hip.so.
#
# NOT EXECUTED in this repository: no `julia` binary exists in the build image (SURVEY.md §0).  What IS checked mechanically
# (tests/test_julia_shim_signatures.py): every `ccall((:name, lib), ret, (types...), ...)` of this file against the prototype
# of `name` in include/lfpsqp_hip.h, every `struct C...` against the C struct of the same role (field order and types), that
# every function the header declares is bound here, and that definitions sit inside the module.  The same C ABI is
# exercised call for call by the Python ctypes mirror (lfpsqp.jl_amd/_capi.py) in tests/.  See INTEGRATION.md.
#
# Layout:  1. raw bindings (one thin wrapper per C entry point)      2. device arrays and BLAS-1/2 dispatch (mul!, dot, norm)
#          3. operators: DiagOperator, DeviceBasis, bound decomposition      4. projcg!, retract! (NR, ProjPenalty, ...), pcg!
#          5. armijo! / exact_linesearch!      6. optimize: the six methods of src/optimize.jl:13-119 on device-resident state
#
# Usage inside LFPSQP.jl:   include("LFPSQPHip.jl"); using .LFPSQPHip
#   ctx  = HipContext(0)
#   prob = QuadLinearBallBox(ctx, n, m, Jct_dev, b; R2 = n / 2, xl = xl, xu = xu)       # device-resident f, c!, jac!, Hessian
#   x, obj_values, λ_kkt, info = optimize(prob, x0, LFPSQPParams())
# or, with arbitrary Julia callables (every user call costs one n-vector PCIe round trip):
#   x, obj_values, λ_kkt, info = optimize(ctx, f, grad!, c!, jac!, hess_lag_vec!, x0, xl, xu, m, LFPSQPParams())
module LFPSQPHip

using LinearAlgebra
using Printf
import LinearAlgebra: mul!, dot, norm

const lib = get(ENV, "LFPSQP_HIP_LIB", joinpath(@__DIR__, "..", "lfpsqp.jl_amd", "lib", "liblfpsqp_hip.so"))

struct HipError <: Exception
    code::Cint
    msg::String
end

# =====================================================================================================================
# 1. C structs of include/lfpsqp_hip.h (isbits mirrors, passed by reference) and raw bindings
# =====================================================================================================================
struct CDiagOp                 # lfpsqp_diag_op
    a0::Float64
    dg::Ptr{Cvoid}
end
struct CLowRankOp              # lfpsqp_lowrank_op
    a0::Float64
    dg::Ptr{Cvoid}
    V::Ptr{Cvoid}
    k::Int64
    sigma::Ptr{Float64}
end
struct CTridiagOp              # lfpsqp_tridiag_op
    a0::Float64
    dg::Ptr{Cvoid}
    off::Ptr{Cvoid}
end
struct CBasis                  # lfpsqp_basis
    Z::Ptr{Cvoid}
    ncols::Int64
    Dx::Ptr{Cvoid}
    Dy::Ptr{Cvoid}
    sx::Ptr{Cvoid}
    sy::Ptr{Cvoid}
    A::Ptr{Cvoid}
    W::Ptr{Float64}
    S::Ptr{Cvoid}               # optional sparse form of Z (lfpsqp_spmat), C_NULL otherwise
    SA::Ptr{Cvoid}              # optional sparse twin of the generator A (projcg in factored form on the nonzeros), C_NULL otherwise
end
struct CWork                   # lfpsqp_projcg_work
    g::Ptr{Cvoid}
    d::Ptr{Cvoid}
    rp::Ptr{Cvoid}
    Utr::Ptr{Cvoid}
end
struct CIneqData               # lfpsqp_ineq_data
    q::Ptr{Cvoid}
    r::Ptr{Cvoid}
    s::Ptr{Cvoid}
    t::Ptr{Cvoid}
    n::Int64
end
struct CElementwise            # lfpsqp_elementwise
    A::Ptr{Cvoid}
    Asp::Ptr{Cvoid}
    kind::Ptr{Cvoid}
    qw::Ptr{Float64}
    work::Ptr{Cvoid}
end
struct CConstraints            # lfpsqp_constraints
    Jct::Ptr{Cvoid}
    m_lin::Int64
    b::Ptr{Float64}
    has_ball::Cint
    R2::Float64
    n_x::Int64
    slack_row::Int64
    Jsp::Ptr{Cvoid}             # optional sparse form of Jct[:, 1:m_lin] (lfpsqp_spmat), C_NULL otherwise
    ew::Ptr{CElementwise}       # optional: the first m_lin constraints are elementwise-transformed linear (nonlinear class), C_NULL otherwise
end
struct CPPWork                 # lfpsqp_pp_work
    r::Ptr{Cvoid}
    p::Ptr{Cvoid}
    z::Ptr{Cvoid}
    dx::Ptr{Cvoid}
    g::Ptr{Cvoid}
    tmp_m::Ptr{Cvoid}
    tmp_w::Ptr{Cvoid}
    h::Ptr{Cvoid}
    DxS::Ptr{Cvoid}
    DyS::Ptr{Cvoid}
    ones::Ptr{Cvoid}
    zeros::Ptr{Cvoid}
    q::Ptr{Cvoid}              # optional (exact preconditioner of the inner solves, DeviceOptions.pp_precondition): work n-vector
    i11::Ptr{Cvoid}            # ... and, with bounds, the rows of D0^-1 (N-vectors)
    i12::Ptr{Cvoid}
    i22::Ptr{Cvoid}
    precondition::Cint
end
struct CPcgPrecond             # lfpsqp_pcg_precond
    K::Ptr{Float64}
    i11::Ptr{Cvoid}
    i12::Ptr{Cvoid}
    i22::Ptr{Cvoid}
    q::Ptr{Cvoid}
end

const LFPSQP_PROJCG_WANT_LAMBDA = Cint(1)
const LFPSQP_PROJCG_RESUME = Cint(2)
const LFPSQP_PROJCG_START_GIVEN = Cint(4)      # work.rp = r0 = -b and work.Utr = U'r0 are given (lfpsqp_tangent_step left them): no residual pass
const LFPSQP_PROJCG_START_PROJECTED = Cint(8)  # the initial projection is done (lfpsqp_tangent_step with LFPSQP_TANGENT_INIT_PROJCG): start with the first iteration
const LFPSQP_TANGENT_INIT_PROJCG = Cint(1)
const LFPSQP_ERR_UNSUPPORTED = Cint(-5)

const H = Ptr{Cvoid}           # an opaque handle (lfpsqp_ctx*, lfpsqp_vec*, lfpsqp_mat*)

# ---- context ---------------------------------------------------------------------------------------------------------
c_ctx_create(device, out) = ccall((:lfpsqp_ctx_create, lib), Cint, (Cint, Ref{Ptr{Cvoid}}), device, out)
c_ctx_destroy(ctx) = ccall((:lfpsqp_ctx_destroy, lib), Cint, (Ptr{Cvoid},), ctx)
c_ctx_sync(ctx) = ccall((:lfpsqp_ctx_sync, lib), Cint, (Ptr{Cvoid},), ctx)
c_last_error(ctx) = ccall((:lfpsqp_last_error, lib), Cstring, (Ptr{Cvoid},), ctx)
c_device_name(ctx, buf, len) = ccall((:lfpsqp_device_name, lib), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Int64), ctx, buf, len)
c_device_uuid(ctx, buf, len) = ccall((:lfpsqp_device_uuid, lib), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Int64), ctx, buf, len)
c_ctx_set_tuning(ctx, ks, nt) = ccall((:lfpsqp_ctx_set_tuning, lib), Cint, (Ptr{Cvoid}, Cint, Cint), ctx, ks, nt)
c_ctx_set_onepass(ctx, mode) = ccall((:lfpsqp_ctx_set_onepass, lib), Cint, (Ptr{Cvoid}, Cint), ctx, mode)
c_ctx_set_residual_buffers(ctx, mode) = ccall((:lfpsqp_ctx_set_residual_buffers, lib), Cint, (Ptr{Cvoid}, Cint), ctx, mode)
c_ctx_stream(ctx, out) = ccall((:lfpsqp_ctx_stream, lib), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), ctx, out)
c_timer_begin(ctx) = ccall((:lfpsqp_timer_begin, lib), Cint, (Ptr{Cvoid},), ctx)
c_timer_end(ctx, ms) = ccall((:lfpsqp_timer_end, lib), Cint, (Ptr{Cvoid}, Ref{Float64}), ctx, ms)
c_ctx_set_profiling(ctx, on) = ccall((:lfpsqp_ctx_set_profiling, lib), Cint, (Ptr{Cvoid}, Cint), ctx, on)
c_profile_read(ctx, ms, counts) = ccall((:lfpsqp_profile_read, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int64}), ctx, ms, counts)
# ---- multi-GPU -----------------------------------------------------------------------------------------------------------
c_shard_range(n, rank, nranks, r0, r1) = ccall((:lfpsqp_shard_range, lib), Cint, (Int64, Cint, Cint, Ref{Int64}, Ref{Int64}), n, rank, nranks, r0, r1)
c_comm_unique_id(ctx, id) = ccall((:lfpsqp_comm_unique_id, lib), Cint, (Ptr{Cvoid}, Ptr{UInt8}), ctx, id)
c_comm_init_rccl(ctx, rank, nranks, id) = ccall((:lfpsqp_comm_init_rccl, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{UInt8}), ctx, rank, nranks, id)
c_comm_init_callback(ctx, rank, nranks, fn, user) = ccall((:lfpsqp_comm_init_callback, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}), ctx, rank, nranks, fn, user)
c_comm_info(ctx, rank, nranks) = ccall((:lfpsqp_comm_info, lib), Cint, (Ptr{Cvoid}, Ref{Cint}, Ref{Cint}), ctx, rank, nranks)
c_comm_p2p_export(ctx, handle) = ccall((:lfpsqp_comm_p2p_export, lib), Cint, (Ptr{Cvoid}, Ptr{UInt8}), ctx, handle)
c_comm_init_p2p(ctx, rank, nranks, handles) = ccall((:lfpsqp_comm_init_p2p, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{UInt8}), ctx, rank, nranks, handles)
c_comm_p2p_allow_coarse(ctx, allow) = ccall((:lfpsqp_comm_p2p_allow_coarse, lib), Cint, (Ptr{Cvoid}, Cint), ctx, allow)
c_comm_p2p_info(ctx, kind, count) = ccall((:lfpsqp_comm_p2p_info, lib), Cint, (Ptr{Cvoid}, Ref{Cint}, Ref{Culonglong}), ctx, kind, count)
# ---- buffers -----------------------------------------------------------------------------------------------------------
c_vec_alloc(ctx, n, out) = ccall((:lfpsqp_vec_alloc, lib), Cint, (Ptr{Cvoid}, Int64, Ref{Ptr{Cvoid}}), ctx, n, out)
c_vec_free(ctx, v) = ccall((:lfpsqp_vec_free, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx, v)
c_vec_len(v) = ccall((:lfpsqp_vec_len, lib), Int64, (Ptr{Cvoid},), v)
c_vec_upload(ctx, v, off, host, count) = ccall((:lfpsqp_vec_upload, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Float64}, Int64), ctx, v, off, host, count)
c_vec_download(ctx, v, off, host, count) = ccall((:lfpsqp_vec_download, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Float64}, Int64), ctx, v, off, host, count)
c_vec_fill(ctx, v, value) = ccall((:lfpsqp_vec_fill, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Float64), ctx, v, value)
c_vec_copy(ctx, dst, src) = ccall((:lfpsqp_vec_copy, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), ctx, dst, src)
c_vec_copy_range(ctx, dst, doff, src, soff, count) = ccall((:lfpsqp_vec_copy_range, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Int64), ctx, dst, doff, src, soff, count)
c_vec_fill_range(ctx, v, off, count, value) = ccall((:lfpsqp_vec_fill_range, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Float64), ctx, v, off, count, value)
c_vec_hash_fill(ctx, v, seed, off, scale, shift) = ccall((:lfpsqp_vec_hash_fill, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, UInt64, Int64, Float64, Float64), ctx, v, seed, off, scale, shift)
c_mat_alloc(ctx, n, m, out) = ccall((:lfpsqp_mat_alloc, lib), Cint, (Ptr{Cvoid}, Int64, Int64, Ref{Ptr{Cvoid}}), ctx, n, m, out)
# placement-tuned allocation (FINDINGS.md 6): candidate allocations tried with the fused projected-CG kernel, the fastest kept
c_ctx_set_placement(ctx, tries) = ccall((:lfpsqp_ctx_set_placement, lib), Cint, (Ptr{Cvoid}, Cint), ctx, tries)
c_ctx_set_nr_batch_mode(ctx, mode) = ccall((:lfpsqp_ctx_set_nr_batch_mode, lib), Cint, (Ptr{Cvoid}, Cint), ctx, mode)
c_mat_alloc_placed(ctx, n, m, out) = ccall((:lfpsqp_mat_alloc_placed, lib), Cint, (Ptr{Cvoid}, Int64, Int64, Ref{Ptr{Cvoid}}), ctx, n, m, out)
c_vecs_alloc_placed(ctx, M, ncols, n, count, out) = ccall((:lfpsqp_vecs_alloc_placed, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Cint, Ptr{Ptr{Cvoid}}), ctx, M, ncols, n, count, out)
c_basis_work_alloc_placed(ctx, n, m, nvec, count, M, out) = ccall((:lfpsqp_basis_work_alloc_placed, lib), Cint, (Ptr{Cvoid}, Int64, Int64, Int64, Cint, Ref{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}), ctx, n, m, nvec, count, M, out)
c_placement_info(ctx, tries, picked, ms, cap) = ccall((:lfpsqp_placement_info, lib), Cint, (Ptr{Cvoid}, Ref{Cint}, Ref{Cint}, Ptr{Float64}, Cint), ctx, tries, picked, ms, cap)
c_placement_probe(ctx, M, ncols, g, d, a, reps, ms) = ccall((:lfpsqp_placement_probe, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ref{Float64}), ctx, M, ncols, g, d, a, reps, ms)
c_mat_free(ctx, M) = ccall((:lfpsqp_mat_free, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx, M)
c_mat_shape(M, n, m) = ccall((:lfpsqp_mat_shape, lib), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}), M, n, m)
c_mat_upload(ctx, M, col0, ncols, host, ldh) = ccall((:lfpsqp_mat_upload, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Int64), ctx, M, col0, ncols, host, ldh)
c_mat_download(ctx, M, col0, ncols, host, ldh) = ccall((:lfpsqp_mat_download, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Int64), ctx, M, col0, ncols, host, ldh)
c_mat_copy(ctx, dst, src) = ccall((:lfpsqp_mat_copy, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), ctx, dst, src)
c_mat_rowscaled_view(ctx, A, rs, out) = ccall((:lfpsqp_mat_rowscaled_view, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), ctx, A, rs, out)
c_mat_view(ctx, A, rs, u, w, out) = ccall((:lfpsqp_mat_view, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), ctx, A, rs, u, w, out)
c_mat_hash_fill(ctx, M, seed, row0, nglob, scale, nrows, ncols) = ccall((:lfpsqp_mat_hash_fill, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, UInt64, Int64, Int64, Float64, Int64, Int64), ctx, M, seed, row0, nglob, scale, nrows, ncols)
# ---- BLAS-1/2 ------------------------------------------------------------------------------------------------------------
c_gemv_t(ctx, M, ncols, v, t) = ccall((:lfpsqp_gemv_t, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}), ctx, M, ncols, v, t)
c_gemv_n(ctx, M, ncols, a, t, b, y) = ccall((:lfpsqp_gemv_n, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Float64, Ptr{Cvoid}, Float64, Ptr{Cvoid}), ctx, M, ncols, a, t, b, y)
c_dot(ctx, x, y, out) = ccall((:lfpsqp_dot, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}), ctx, x, y, out)
c_dot_head(ctx, x, y, count, out) = ccall((:lfpsqp_dot_head, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ref{Float64}), ctx, x, y, count, out)
c_nrm2(ctx, x, out) = ccall((:lfpsqp_nrm2, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}), ctx, x, out)
c_amax(ctx, x, out) = ccall((:lfpsqp_amax, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}), ctx, x, out)
c_axpby(ctx, a, x, b, y) = ccall((:lfpsqp_axpby, lib), Cint, (Ptr{Cvoid}, Float64, Ptr{Cvoid}, Float64, Ptr{Cvoid}), ctx, a, x, b, y)
c_waxpby(ctx, a, x, b, y, z) = ccall((:lfpsqp_waxpby, lib), Cint, (Ptr{Cvoid}, Float64, Ptr{Cvoid}, Float64, Ptr{Cvoid}, Ptr{Cvoid}), ctx, a, x, b, y, z)
c_vmul(ctx, d, x, y) = ccall((:lfpsqp_vmul, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), ctx, d, x, y)
c_affine_head(ctx, a, x, c, count, y) = ccall((:lfpsqp_affine_head, lib), Cint, (Ptr{Cvoid}, Float64, Ptr{Cvoid}, Float64, Int64, Ptr{Cvoid}), ctx, a, x, c, count, y)
c_sumsq_shift(ctx, x, count, c, out) = ccall((:lfpsqp_sumsq_shift, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Float64, Ref{Float64}), ctx, x, count, c, out)
c_separable(ctx, kind, mode, a, a0, c, c0, x, count, outv, outs) = ccall((:lfpsqp_separable, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Float64, Ptr{Cvoid}, Float64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Float64}), ctx, kind, mode, a, a0, c, c0, x, count, outv, outs)
c_allreduce(ctx, v, count) = ccall((:lfpsqp_allreduce, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), ctx, v, count)
# ---- sparse constraint gradients ---------------------------------------------------------------------------------------------
c_spmat_create(ctx, n, m, nnz, rows, cols, vals, out) = ccall((:lfpsqp_spmat_create, lib), Cint, (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ref{Ptr{Cvoid}}), ctx, n, m, nnz, rows, cols, vals, out)
c_spmat_free(ctx, S) = ccall((:lfpsqp_spmat_free, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx, S)
c_spmat_info(S, n, m, nnz, k) = ccall((:lfpsqp_spmat_info, lib), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ref{Int64}, Ref{Int64}), S, n, m, nnz, k)
c_spmv_t(ctx, S, v, t) = ccall((:lfpsqp_spmv_t, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), ctx, S, v, t)
c_spmv_n(ctx, S, a, t, b, y) = ccall((:lfpsqp_spmv_n, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Float64, Ptr{Cvoid}, Float64, Ptr{Cvoid}), ctx, S, a, t, b, y)
c_spmat_to_dense(ctx, S, M) = ccall((:lfpsqp_spmat_to_dense, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), ctx, S, M)
c_spmat_clone(ctx, S, out) = ccall((:lfpsqp_spmat_clone, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), ctx, S, out)
c_spmat_rowscale(ctx, dst, src, v) = ccall((:lfpsqp_spmat_rowscale, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), ctx, dst, src, v)
c_spmat_gram(ctx, S, Jct, w2, G) = ccall((:lfpsqp_spmat_gram, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}), ctx, S, Jct, w2, G)
# ---- bound manifolds -------------------------------------------------------------------------------------------------------
c_half_stride(N) = ccall((:lfpsqp_half_stride, lib), Int64, (Int64,), N)
c_ineq_data_build(ctx, xl, xu, q, r, s, t) = ccall((:lfpsqp_ineq_data_build, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), ctx, xl, xu, q, r, s, t)
c_generate_initial_y(ctx, xaug, id) = ccall((:lfpsqp_generate_initial_y, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{CIneqData}), ctx, xaug, id)
c_calculate_h(ctx, h, xaug, id, hmax) = ccall((:lfpsqp_calculate_h, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{CIneqData}, Ptr{Float64}), ctx, h, xaug, id, hmax)
c_inequality_gradient(ctx, xaug, id, Dx, Dy, S, sx, sy) = ccall((:lfpsqp_inequality_gradient, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{CIneqData}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), ctx, xaug, id, Dx, Dy, S, sx, sy)
c_calculate_lambda_y(ctx, Jct, ncols, lam, Dx, S, w, lamy) = ccall((:lfpsqp_calculate_lambda_y, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), ctx, Jct, ncols, lam, Dx, S, w, lamy)
c_augmented_diag(ctx, hx, lamy, id, a) = ccall((:lfpsqp_augmented_diag, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{CIneqData}, Ptr{Cvoid}), ctx, hx, lamy, id, a)
c_y_retract(ctx, xnew, x, id) = ccall((:lfpsqp_y_retract, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{CIneqData}), ctx, xnew, x, id)
# ---- tangent setup -----------------------------------------------------------------------------------------------------------
c_gram(ctx, M, ncols, w2, G) = ccall((:lfpsqp_gram, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Float64}), ctx, M, ncols, w2, G)
c_rmul(ctx, In, kcols, W, rcols, Out) = ccall((:lfpsqp_rmul, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Float64}, Int64, Ptr{Cvoid}), ctx, In, kcols, W, rcols, Out)
c_factorize_hint(ctx, Vt_prev, m) = ccall((:lfpsqp_factorize_hint, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64), ctx, Vt_prev, m)
c_factorize(ctx, Jct, w2, Z, Sigma, Vt, W, rank, eps_rank) = ccall((:lfpsqp_factorize, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Int64}, Float64), ctx, Jct, w2, Z, Sigma, Vt, W, rank, eps_rank)
c_gram_rhs(ctx, M, ncols, w2, nx, e, G, X) = ccall((:lfpsqp_gram_rhs, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Ptr{Cvoid}}, Ptr{Float64}, Ptr{Float64}), ctx, M, ncols, w2, nx, e, G, X)
c_factorize_rhs(ctx, Jct, w2, Z, Sigma, Vt, W, rank, eps_rank, e, Jte, G) = ccall((:lfpsqp_factorize_rhs, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Int64}, Float64, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), ctx, Jct, w2, Z, Sigma, Vt, W, rank, eps_rank, e, Jte, G)
c_factorize_sp(ctx, S, Jct, w2, Z, Sigma, Vt, W, rank, eps_rank) = ccall((:lfpsqp_factorize_sp, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Int64}, Float64), ctx, S, Jct, w2, Z, Sigma, Vt, W, rank, eps_rank)
c_small_svd(ctx, rows, cols, A, U, S, V) = ccall((:lfpsqp_small_svd, lib), Cint, (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), ctx, rows, cols, A, U, S, V)
c_q_gemv_t(ctx, Q, v, w, t) = ccall((:lfpsqp_q_gemv_t, lib), Cint, (Ptr{Cvoid}, Ref{CBasis}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), ctx, Q, v, w, t)
c_q_gemv_n(ctx, Q, a, w, t, b, y) = ccall((:lfpsqp_q_gemv_n, lib), Cint, (Ptr{Cvoid}, Ref{CBasis}, Float64, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Ptr{Cvoid}), ctx, Q, a, w, t, b, y)
# ---- solvers -------------------------------------------------------------------------------------------------------------------
c_factored_basis_supported(ctx, A, SA, yes) = ccall((:lfpsqp_factored_basis_supported, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Cint}), ctx, A, SA, yes)
c_projcg(ctx, x, lam, A, U, b, c, tol, maxit, nglob, flags, work, iters, nr) = ccall((:lfpsqp_projcg, lib), Cint,
    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{CDiagOp}, Ref{CBasis}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Int64, Int64, Cint, Ref{CWork}, Ref{Int64}, Ref{Float64}),
    ctx, x, lam, A, U, b, c, tol, maxit, nglob, flags, work, iters, nr)
c_projcg_lowrank(ctx, x, lam, A, U, b, c, tol, maxit, nglob, flags, work, iters, nr) = ccall((:lfpsqp_projcg_lowrank, lib), Cint,
    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{CLowRankOp}, Ref{CBasis}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Int64, Int64, Cint, Ref{CWork}, Ref{Int64}, Ref{Float64}),
    ctx, x, lam, A, U, b, c, tol, maxit, nglob, flags, work, iters, nr)
c_projcg_tridiag(ctx, x, lam, A, Av, U, b, c, tol, maxit, nglob, flags, work, iters, nr) = ccall((:lfpsqp_projcg_tridiag, lib), Cint,
    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{CTridiagOp}, Ptr{Cvoid}, Ref{CBasis}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Int64, Int64, Cint, Ref{CWork}, Ref{Int64}, Ref{Float64}),
    ctx, x, lam, A, Av, U, b, c, tol, maxit, nglob, flags, work, iters, nr)
c_tridiag_mul(ctx, A, v, out) = ccall((:lfpsqp_tridiag_mul, lib), Cint, (Ptr{Cvoid}, Ref{CTridiagOp}, Ptr{Cvoid}, Ptr{Cvoid}), ctx, A, v, out)
c_projcg_op(ctx, x, lam, A, user, Av, U, b, c, tol, maxit, nglob, flags, work, iters, nr) = ccall((:lfpsqp_projcg_op, lib), Cint,
    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{CBasis}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Int64, Int64, Cint, Ref{CWork}, Ref{Int64}, Ref{Float64}),
    ctx, x, lam, A, user, Av, U, b, c, tol, maxit, nglob, flags, work, iters, nr)
c_tangent_step(ctx, U, Sigma, Vt, m, Jtd, G, d, cons, x, hdiag, idata, hx, S, lamy, work, flags, Utd, lam, dss) = ccall((:lfpsqp_tangent_step, lib), Cint,
    (Ptr{Cvoid}, Ref{CBasis}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}, Ptr{CConstraints}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{CIneqData}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
     Ref{CWork}, Cint, Ptr{Float64}, Ptr{Float64}, Ref{Float64}),
    ctx, U, Sigma, Vt, m, Jtd, G, d, cons, x, hdiag, idata, hx, S, lamy, work, flags, Utd, lam, dss)
c_ineq_rhs(ctx, daug, Dx, Dy, e) = ccall((:lfpsqp_ineq_rhs, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), ctx, daug, Dx, Dy, e)
c_constraints_eval(ctx, cons, x, cval) = ccall((:lfpsqp_constraints_eval, lib), Cint, (Ptr{Cvoid}, Ref{CConstraints}, Ptr{Cvoid}, Ptr{Float64}), ctx, cons, x, cval)
c_constraints_jac(ctx, cons, x, Jct, cval) = ccall((:lfpsqp_constraints_jac, lib), Cint, (Ptr{Cvoid}, Ref{CConstraints}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}), ctx, cons, x, Jct, cval)
c_constraints_hess_diag(ctx, cons, x, lam, hx) = ccall((:lfpsqp_constraints_hess_diag, lib), Cint, (Ptr{Cvoid}, Ref{CConstraints}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Cvoid}), ctx, cons, x, lam, hx)
c_retract_nr(ctx, U, Sigma, Vt, m, cons, cfun, cuser, idata, xtilde, x, xnew, tol, maxiter, cval, flag, iters) = ccall((:lfpsqp_retract_nr, lib), Cint,
    (Ptr{Cvoid}, Ref{CBasis}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{CConstraints}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{CIneqData}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Int64, Ptr{Float64}, Ref{Cint}, Ref{Int64}),
    ctx, U, Sigma, Vt, m, cons, cfun, cuser, idata, xtilde, x, xnew, tol, maxiter, cval, flag, iters)
c_retract_nr_batch_width(ctx, U, cons, width) = ccall((:lfpsqp_retract_nr_batch_width, lib), Cint, (Ptr{Cvoid}, Ref{CBasis}, Ref{CConstraints}, Ref{Cint}), ctx, U, cons, width)
c_retract_nr_batch(ctx, U, Sigma, Vt, m, cons, idata, nb, xtilde, x, xnew, tol, maxiter, cval, flags, iters) = ccall((:lfpsqp_retract_nr_batch, lib), Cint,
    (Ptr{Cvoid}, Ref{CBasis}, Ptr{Float64}, Ptr{Float64}, Int64, Ref{CConstraints}, Ptr{CIneqData}, Cint, Ptr{Ptr{Cvoid}}, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Float64, Int64, Ptr{Float64}, Ptr{Cint}, Ptr{Int64}),
    ctx, U, Sigma, Vt, m, cons, idata, nb, xtilde, x, xnew, tol, maxiter, cval, flags, iters)
c_pcg_pre(ctx, mu, J, P, x, r, p, z, tol, maxiter, flag, iters) = ccall((:lfpsqp_pcg_pre, lib), Cint,
    (Ptr{Cvoid}, Float64, Ref{CBasis}, Ref{CPcgPrecond}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Int64, Ref{Cint}, Ref{Int64}),
    ctx, mu, J, P, x, r, p, z, tol, maxiter, flag, iters)
c_pcg(ctx, mu, Jop, x, r, p, z, tmp_w, tmp_m, tol, maxiter, flag, iters) = ccall((:lfpsqp_pcg, lib), Cint,
    (Ptr{Cvoid}, Float64, Ref{CBasis}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Int64, Ref{Cint}, Ref{Int64}),
    ctx, mu, Jop, x, r, p, z, tmp_w, tmp_m, tol, maxiter, flag, iters)
c_retract_pp(ctx, cons, cfun, jacfun, user, Jct, m, idata, Dx, Dy, S, xtilde, x, xnew, mu0, tol, maxiter, maxiter_pcg, work, cval, flag, iters, pcg_iters) = ccall((:lfpsqp_retract_pp, lib), Cint,
    (Ptr{Cvoid}, Ptr{CConstraints}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{CIneqData}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Int64, Int64, Ref{CPPWork}, Ptr{Float64}, Ref{Cint}, Ref{Int64}, Ref{Int64}),
    ctx, cons, cfun, jacfun, user, Jct, m, idata, Dx, Dy, S, xtilde, x, xnew, mu0, tol, maxiter, maxiter_pcg, work, cval, flag, iters, pcg_iters)

# =====================================================================================================================
# 2. Context, device arrays, BLAS-1/2 dispatch
# =====================================================================================================================
# Options of the DEVICE implementation with no counterpart in the reference -- kept out of LFPSQPParams, which mirrors src/LFPSQP.jl:57-81
# field for field.  ls_batch: trial retractions of a failing linesearch that share their passes over Jct (1 = off; 0 = automatic: what the
# previous search's failures suggest, at least 4, up to what a pass takes for the shape and batch mode; k > 1 = at most k);
# ls_batch_matrix_cores: false (default) = the EXACT batch (lfpsqp_ctx_set_nr_batch_mode: up to 4 trials per pass, bit for bit the one-by-one
# retractions -- the batched search IS the reference's search, also where it is chaotic), true = the matrix-core batch (up to 16 per pass,
# equal up to rounding: a chaotic search may accept another step); placement_tries: candidate allocations per placement-tuned buffer (set_placement!).
mutable struct DeviceOptions
    ls_batch::Int
    ls_batch_matrix_cores::Bool
    placement_tries::Int
    factored_basis::Bool        # keep the tangent basis in factored form U = Jct W whenever the fused projected-CG iteration applies
    pp_precondition::Bool       # ProjPenalty's inner solves with the exact preconditioner of their operator (lfpsqp_pcg_pre); false = the reference's live path
    warm_factorize::Bool        # the small eigenproblem of the tangent setup starts from the previous outer iteration's Vt (lfpsqp_factorize_hint)
    fused_tangent_step::Bool    # Jct'd rides with the Gram pass; ONE pass projects d, completes the Hessian diagonal and forms projcg!'s first U'r (lfpsqp_tangent_step)
    tridiagonal_one_pass::Bool  # tridiagonal Lagrangian Hessians (hess_offdiag) on the one-pass solver (lfpsqp_projcg_tridiag); false: through the callback path
end
mutable struct HipContext
    h::Ptr{Cvoid}
    rank::Int
    nranks::Int
    options::DeviceOptions
    function HipContext(device::Integer=0)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        rc = c_ctx_create(Cint(device), r)
        rc == 0 || throw(HipError(rc, "lfpsqp_ctx_create failed: no usable MI355X (there is no CPU fallback)"))
        ctx = new(r[], 0, 1, DeviceOptions(0, false, 3, true, false, true, true, true))
        finalizer(c -> c_ctx_destroy(c.h), ctx)
        return ctx
    end
end
check(ctx::HipContext, rc::Cint) = rc == 0 ? nothing : throw(HipError(rc, unsafe_string(c_last_error(ctx.h))))
sync(ctx::HipContext) = check(ctx, c_ctx_sync(ctx.h))

# multi-GPU: one Julia process per GPU (e.g. under MPI.jl); rank 0 creates the id and broadcasts it
function comm_unique_id(ctx::HipContext)
    id = Vector{UInt8}(undef, 128)
    check(ctx, c_comm_unique_id(ctx.h, id))
    return id
end
# the one-shot peer-to-peer all-reduce (one node): every rank exports its mailbox handle (64 bytes), all handles go to all ranks
# (e.g. MPI.Allgather), comm_init_p2p! maps them -- instead of comm_unique_id / comm_init!
function comm_p2p_export(ctx::HipContext)
    h = Vector{UInt8}(undef, 64)
    check(ctx, c_comm_p2p_export(ctx.h, h))
    return h
end
comm_p2p_allow_coarse!(ctx::HipContext, allow::Bool=true) = (check(ctx, c_comm_p2p_allow_coarse(ctx.h, Cint(allow))); ctx)
# (memory kind of this rank's mailbox -- :fine, :coarse or :none --, all-reduce launches so far)
function comm_p2p_info(ctx::HipContext)
    k = Ref{Cint}(0); cnt = Ref{Culonglong}(0)
    check(ctx, c_comm_p2p_info(ctx.h, k, cnt))
    return ((:none, :fine, :coarse)[k[] + 1], cnt[])
end
function comm_init_p2p!(ctx::HipContext, rank::Integer, nranks::Integer, handles::Vector{UInt8})
    length(handles) == 64 * nranks || error("handles must hold 64 bytes per rank, in rank order")
    check(ctx, c_comm_init_p2p(ctx.h, Cint(rank), Cint(nranks), handles))
    ctx.rank, ctx.nranks = rank, nranks
    return ctx
end
function comm_init!(ctx::HipContext, rank::Integer, nranks::Integer, id::Vector{UInt8})
    check(ctx, c_comm_init_rccl(ctx.h, Cint(rank), Cint(nranks), id))
    ctx.rank, ctx.nranks = rank, nranks
    return ctx
end
function shard_range(n::Integer, rank::Integer, nranks::Integer)
    r0 = Ref{Int64}(0); r1 = Ref{Int64}(0)
    c_shard_range(Int64(n), Cint(rank), Cint(nranks), r0, r1) == 0 || error("lfpsqp_shard_range: invalid arguments")
    return Int(r0[]), Int(r1[])                 # 0-based half-open row range [r0, r1)
end

# A device vector.  With bounds the reference doubles the variables (xaug = [x; y], length 2N); on the device such a STACKED
# vector keeps the x-half at [0, N) and the y-half at [hs, hs + N) (hs = lfpsqp_half_stride(N)): N > 0 marks it.
mutable struct DeviceVector <: AbstractVector{Float64}
    ctx::HipContext
    h::Ptr{Cvoid}
    n::Int                      # allocated logical length (hs + N when stacked)
    N::Int                      # 0: plain vector; > 0: stacked [x | gap | y] of a 2N-vector
    hs::Int
end
function DeviceVector(ctx::HipContext, n::Integer)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ctx, c_vec_alloc(ctx.h, Int64(n), r))
    v = DeviceVector(ctx, r[], n, 0, 0)
    finalizer(x -> c_vec_free(x.ctx.h, x.h), v)
    return v
end
function StackedVector(ctx::HipContext, N::Integer)
    hs = Int(c_half_stride(Int64(N)))
    v = DeviceVector(ctx, hs + N)
    v.N, v.hs = N, hs
    return v
end
similar_device(v::DeviceVector) = v.N > 0 ? StackedVector(v.ctx, v.N) : DeviceVector(v.ctx, v.n)
Base.size(v::DeviceVector) = (v.N > 0 ? 2 * v.N : v.n,)
Base.length(v::DeviceVector) = v.N > 0 ? 2 * v.N : v.n
function upload!(v::DeviceVector, host::AbstractVector{Float64}, offset::Integer=0)
    hostc = Vector{Float64}(host)
    check(v.ctx, c_vec_upload(v.ctx.h, v.h, Int64(offset), hostc, Int64(length(hostc))))
    return v
end
function download(v::DeviceVector, count::Integer=v.n, offset::Integer=0)
    host = Vector{Float64}(undef, count)
    check(v.ctx, c_vec_download(v.ctx.h, v.h, Int64(offset), host, Int64(count)))
    return host
end
upload2!(v::DeviceVector, host::AbstractVector{Float64}) = (upload!(v, host[1:v.N], 0); upload!(v, host[v.N+1:2*v.N], v.hs); v)
download2(v::DeviceVector) = vcat(download(v, v.N, 0), download(v, v.N, v.hs))
Base.fill!(v::DeviceVector, value::Real) = (check(v.ctx, c_vec_fill(v.ctx.h, v.h, Float64(value))); v)
Base.copyto!(dst::DeviceVector, src::DeviceVector) = (check(dst.ctx, c_vec_copy(dst.ctx.h, dst.h, src.h)); dst)
copy_range!(dst::DeviceVector, doff::Integer, src::DeviceVector, soff::Integer, count::Integer) =
    (check(dst.ctx, c_vec_copy_range(dst.ctx.h, dst.h, Int64(doff), src.h, Int64(soff), Int64(count))); dst)
fill_range!(v::DeviceVector, off::Integer, count::Integer, value::Real) =
    (check(v.ctx, c_vec_fill_range(v.ctx.h, v.h, Int64(off), Int64(count), Float64(value))); v)

mutable struct DeviceMatrix <: AbstractMatrix{Float64}
    ctx::HipContext
    h::Ptr{Cvoid}
    n::Int
    m::Int
end
function DeviceMatrix(ctx::HipContext, n::Integer, m::Integer; placed::Bool=false)      # placed: allocate by trial (lfpsqp_mat_alloc_placed)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ctx, placed ? c_mat_alloc_placed(ctx.h, Int64(n), Int64(m), r) : c_mat_alloc(ctx.h, Int64(n), Int64(m), r))
    M = DeviceMatrix(ctx, r[], n, m)
    finalizer(x -> c_mat_free(x.ctx.h, x.h), M)
    return M
end
Base.size(M::DeviceMatrix) = (M.n, M.m)
# Diagonal(rs) * A + u * w' without a copy (lfpsqp_mat_view; any part may be nothing): accepted wherever the library only reads a matrix through
# its product kernels; VIEW_KEEP pins the borrowed storage and vectors for the view's lifetime
const VIEW_KEEP = IdDict{Any,Any}()
function matrix_view(A::DeviceMatrix; rs::Union{Nothing,DeviceVector}=nothing, u::Union{Nothing,DeviceVector}=nothing, w::Union{Nothing,DeviceVector}=nothing)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    hp(x) = x === nothing ? C_NULL : x.h
    check(A.ctx, c_mat_view(A.ctx.h, A.h, hp(rs), hp(u), hp(w), r))
    V = DeviceMatrix(A.ctx, r[], A.n, A.m)
    VIEW_KEEP[V] = (A, rs, u, w)
    finalizer(x -> (c_mat_free(x.ctx.h, x.h); delete!(VIEW_KEEP, x)), V)
    return V
end
function rowscaled_view(A::DeviceMatrix, rs::DeviceVector)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(A.ctx, c_mat_rowscaled_view(A.ctx.h, A.h, rs.h, r))
    V = DeviceMatrix(A.ctx, r[], A.n, A.m)
    VIEW_KEEP[V] = (A, rs)
    finalizer(x -> (c_mat_free(x.ctx.h, x.h); delete!(VIEW_KEEP, x)), V)
    return V
end
set_nr_batch_mode!(ctx::HipContext, matrix_cores::Bool) = (check(ctx, c_ctx_set_nr_batch_mode(ctx.h, Cint(matrix_cores ? 1 : 0))); ctx.options.ls_batch_matrix_cores = matrix_cores; ctx)      # default: the exact batch
set_placement!(ctx::HipContext, tries::Integer) = (check(ctx, c_ctx_set_placement(ctx.h, Cint(tries))); ctx.options.placement_tries = tries; ctx)      # 1 = off, default 3
# The basis (n x m) and `count` n-vectors streamed with it (stacked [x | gap | y] vectors of 2N entries when N > 0), allocated TOGETHER by
# trial over every pair of candidate allocations -- the speed of the fused projected-CG kernel is a property of the PAIR
# (lfpsqp_basis_work_alloc_placed).  Returns (DeviceMatrix, Vector{DeviceVector}); the vectors share one allocation.
function basis_and_vectors_placed(ctx::HipContext, n::Integer, m::Integer, count::Integer; N::Integer=0)
    hs = N > 0 ? Int(c_half_stride(Int64(N))) : 0
    nv = N > 0 ? hs + N : n
    hh = fill(Ptr{Cvoid}(C_NULL), count)
    mr = Ref{Ptr{Cvoid}}(C_NULL)
    check(ctx, c_basis_work_alloc_placed(ctx.h, Int64(n), Int64(m), Int64(nv), Cint(count), mr, hh))
    M = DeviceMatrix(ctx, mr[], n, m)
    finalizer(x -> c_mat_free(x.ctx.h, x.h), M)
    vs = DeviceVector[]
    for h in hh
        v = DeviceVector(ctx, h, nv, N, hs)
        finalizer(x -> c_vec_free(x.ctx.h, x.h), v)
        push!(vs, v)
    end
    return M, vs
end
# `count` n-vectors (stacked when N > 0) from ONE allocation, placement-tuned against the matrix M they are streamed with (lfpsqp_vecs_alloc_placed)
function vectors_placed(ctx::HipContext, M::DeviceMatrix, n::Integer, count::Integer; N::Integer=0)
    hs = N > 0 ? Int(c_half_stride(Int64(N))) : 0
    nv = N > 0 ? hs + N : n
    hh = fill(Ptr{Cvoid}(C_NULL), count)
    check(ctx, c_vecs_alloc_placed(ctx.h, M.h, Int64(M.m), Int64(nv), Cint(count), hh))
    vs = DeviceVector[]
    for h in hh
        v = DeviceVector(ctx, h, nv, N, hs)
        finalizer(x -> c_vec_free(x.ctx.h, x.h), v)
        push!(vs, v)
    end
    return vs
end
function placement_info(ctx::HipContext)                    # (trials made, index kept, fused-kernel ms per trial) of the last placed allocation
    tries, picked = Ref{Cint}(0), Ref{Cint}(0)
    ms = zeros(64)
    check(ctx, c_placement_info(ctx.h, tries, picked, ms, Cint(64)))
    return Int(tries[]), Int(picked[]), ms[1:tries[]]
end
function upload!(M::DeviceMatrix, host::Matrix{Float64}, col0::Integer=0)      # Julia matrices are column-major: the layouts match
    check(M.ctx, c_mat_upload(M.ctx.h, M.h, Int64(col0), Int64(size(host, 2)), host, Int64(max(size(host, 1), 1))))
    return M
end
function download(M::DeviceMatrix)
    host = Matrix{Float64}(undef, M.n, M.m)
    check(M.ctx, c_mat_download(M.ctx.h, M.h, Int64(0), Int64(M.m), host, Int64(max(M.n, 1))))
    return host
end

# sparse n_loc x m constraint gradients with a few nonzeros per row (lfpsqp_spmat), from 1-based Julia triplets / findnz(sparse(A))
mutable struct SparseMatrix
    ctx::HipContext
    h::Ptr{Cvoid}
    n::Int
    m::Int
end
function SparseMatrix(ctx::HipContext, n::Integer, m::Integer, I::Vector{Int}, J::Vector{Int}, V::Vector{Float64})
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ctx, c_spmat_create(ctx.h, Int64(n), Int64(m), Int64(length(V)), Int64.(I .- 1), Int64.(J .- 1), V, r))
    S = SparseMatrix(ctx, r[], n, m)
    finalizer(x -> c_spmat_free(x.ctx.h, x.h), S)
    return S
end
spmv_t!(t::DeviceVector, S::SparseMatrix, v::DeviceVector) = (check(t.ctx, c_spmv_t(t.ctx.h, S.h, v.h, t.h)); t)                 # t = S'v
spmv_n!(y::DeviceVector, S::SparseMatrix, t::DeviceVector, a::Real=1.0, b::Real=0.0) =
    (check(y.ctx, c_spmv_n(y.ctx.h, S.h, Float64(a), t.h, Float64(b), y.h)); y)                                                 # y = a S t + b y
to_dense!(M::DeviceMatrix, S::SparseMatrix) = (check(M.ctx, c_spmat_to_dense(M.ctx.h, S.h, M.h)); M)
# a second object with S's STRUCTURE (shared on the device: keep S alive) and its own values -- the x-dependent constraint gradients
# diag(phi'(x)) A of ElementwiseConstraints, rescaled in place by rowscale!
function clone(S::SparseMatrix)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(S.ctx, c_spmat_clone(S.ctx.h, S.h, r))
    C = SparseMatrix(S.ctx, r[], S.n, S.m)
    finalizer(x -> c_spmat_free(x.ctx.h, x.h), C)
    return C
end
rowscale!(dst::SparseMatrix, src::SparseMatrix, v::DeviceVector) = (check(dst.ctx, c_spmat_rowscale(dst.ctx.h, dst.h, src.h, v.h)); dst)   # dst.values = Diagonal(v) * src.values
# [S | Jct[:, m+1:end]]' * Diagonal(w2) * [S | ...] from the nonzeros, exactly accumulated (order-independent)
function gram(S::SparseMatrix; Jct::Union{DeviceMatrix,Nothing}=nothing, w2::Union{DeviceVector,Nothing}=nothing)
    M = Jct === nothing ? S.m : Jct.m
    G = zeros(Float64, M, M)
    check(S.ctx, c_spmat_gram(S.ctx.h, S.h, Jct === nothing ? C_NULL : Jct.h, w2 === nothing ? C_NULL : w2.h, G))
    return G
end

# BLAS-1 on device vectors (reductions are global: all-reduced over the ranks and replicated)
function dot(x::DeviceVector, y::DeviceVector)
    r = Ref{Float64}(0.0)
    check(x.ctx, c_dot(x.ctx.h, x.h, y.h, r))
    return r[]
end
function norm(x::DeviceVector, p::Real=2)
    r = Ref{Float64}(0.0)
    check(x.ctx, p == Inf ? c_amax(x.ctx.h, x.h, r) : c_nrm2(x.ctx.h, x.h, r))
    return r[]
end
function norm_head(x::DeviceVector, count::Integer)        # norm(view(step, 1:n)), src/linesearch.jl:66
    r = Ref{Float64}(0.0)
    check(x.ctx, c_dot_head(x.ctx.h, x.h, x.h, Int64(count), r))
    return sqrt(r[])
end
axpby!(a::Real, x::DeviceVector, b::Real, y::DeviceVector) = (check(y.ctx, c_axpby(y.ctx.h, Float64(a), x.h, Float64(b), y.h)); y)
waxpby!(z::DeviceVector, a::Real, x::DeviceVector, b::Real, y::DeviceVector) =
    (check(z.ctx, c_waxpby(z.ctx.h, Float64(a), x.h, Float64(b), y.h, z.h)); z)            # z = a x + b y
vmul!(y::DeviceVector, d::DeviceVector, x::DeviceVector) = (check(y.ctx, c_vmul(y.ctx.h, d.h, x.h, y.h)); y)

# =====================================================================================================================
# 3. Operators
# =====================================================================================================================
# view(U, :, 1:rank) (src/optimize.jl:370).  `generator` = (Jct, W) with Z == Jct*W (ksvd!'s W): the Newton retraction then
# streams Jct once per step instead of Z and Jct.
# Z === nothing with generator = (A, W): the basis in FACTORED form U = A W, never materialised -- projcg!, the projections and the Newton
# retraction stream A and apply the small factor W on the side (lfpsqp_basis.Z == NULL, FINDINGS.md 5.3).
struct DeviceBasis
    Z::Union{Nothing,DeviceMatrix}
    ncols::Int
    generator::Union{Nothing,Tuple{DeviceMatrix,Matrix{Float64}}}
    sparse::Ptr{Cvoid}         # lfpsqp_spmat handle of the generator's sparse twin (projcg on the nonzeros), C_NULL otherwise
end
DeviceBasis(Z::DeviceMatrix) = DeviceBasis(Z, Z.m, nothing, C_NULL)
DeviceBasis(Z::DeviceMatrix, ncols::Integer) = DeviceBasis(Z, ncols, nothing, C_NULL)
DeviceBasis(Z::Union{Nothing,DeviceMatrix}, ncols::Integer, generator) = DeviceBasis(Z, ncols, generator, C_NULL)
zhandle(Z) = Z === nothing ? Ptr{Cvoid}(C_NULL) : Z.h
struct DeviceBasisAdjoint
    U::DeviceBasis
end
Base.adjoint(U::DeviceBasis) = DeviceBasisAdjoint(U)
Base.adjoint(Ut::DeviceBasisAdjoint) = Ut.U
# kgemv! (src/la_helper.jl:36-44) and the mul! calls of src/projcg.jl / src/retractions.jl
# can projcg! run on a basis kept in factored form U = A W (no Z) on this context?  (lfpsqp_factored_basis_supported)
function factored_basis_supported(ctx::HipContext, A::DeviceMatrix, SA::Ptr{Cvoid}=C_NULL)
    yes = Ref{Cint}(0)
    check(ctx, c_factored_basis_supported(ctx.h, A.h, SA, yes))
    return yes[] != 0
end
# (a basis that carries its generator and the generator's sparse twin is applied in factored form on the nonzeros: lfpsqp_q_gemv_*)
factored(U::DeviceBasis) = U.generator !== nothing && (U.sparse != C_NULL || U.Z === nothing)
mul!(y::DeviceVector, U::DeviceBasis, t::DeviceVector, a::Number=1.0, b::Number=0.0) =
    (check(y.ctx, factored(U) ? c_q_gemv_n(y.ctx.h, Ref(cbasis(U)), Float64(a), C_NULL, t.h, Float64(b), y.h) :
                                c_gemv_n(y.ctx.h, U.Z.h, Int64(U.ncols), Float64(a), t.h, Float64(b), y.h)); y)
mul!(t::DeviceVector, Ut::DeviceBasisAdjoint, v::DeviceVector) =
    (check(t.ctx, factored(Ut.U) ? c_q_gemv_t(t.ctx.h, Ref(cbasis(Ut.U)), v.h, C_NULL, t.h) :
                                   c_gemv_t(t.ctx.h, Ut.U.Z.h, Int64(Ut.U.ncols), v.h, t.h)); t)
cbasis(U::DeviceBasis) = U.generator === nothing ? CBasis(U.Z.h, U.ncols, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL) :
    CBasis(zhandle(U.Z), U.ncols, C_NULL, C_NULL, C_NULL, C_NULL, U.generator[1].h, pointer(U.generator[2]), C_NULL, U.sparse)

# A = a0*I + diag(dg): the LinearMap of src/optimize.jl:228-230 for diagonal Lagrangian Hessians
struct DiagOperator
    a0::Float64
    dg::Union{Nothing,DeviceVector}
end
DiagOperator(a0::Real) = DiagOperator(Float64(a0), nothing)
# A = a0*I + Diagonal(dg) + V*Diagonal(σ)*V' (k <= 8 columns of the device matrix V): a diagonal Hessian with a few coupling directions.
# projcg! runs it on the fused ONE-pass iteration (lfpsqp_projcg_lowrank) where a LinearMap closure would take two passes per iteration.
struct LowRankOperator
    a0::Float64
    dg::Union{Nothing,DeviceVector}
    V::DeviceMatrix
    k::Int
    σ::Vector{Float64}
end
LowRankOperator(a0::Real, dg, V::DeviceMatrix; k::Int=V.m, σ::Vector{Float64}=ones(k)) = LowRankOperator(Float64(a0), dg, V, k, σ)

# (A v)_i = (a0 + dg_i) v_i + off_{i-1} v_{i-1} + off_i v_{i+1}: a diagonal Hessian with nearest-neighbour couplings (off: length n, its last entry
# is ignored).  projcg! keeps ONE pass over the basis per iteration with it (lfpsqp_projcg_tridiag); mul! is the LinearMap's action.
struct TridiagonalOperator
    a0::Float64
    dg::Union{Nothing,DeviceVector}
    off::DeviceVector
end
ctridiag(A::TridiagonalOperator) = CTridiagOp(A.a0, A.dg === nothing ? C_NULL : A.dg.h, A.off.h)
function LinearAlgebra.mul!(dest::DeviceVector, A::TridiagonalOperator, v::DeviceVector)
    GC.@preserve A check(dest.ctx, c_tridiag_mul(dest.ctx.h, Ref(ctridiag(A)), v.h, dest.h))
    return dest
end

# InequalityData(xl, xu) (src/inequality_helper.jl:39-89), device-resident q, r, s, t
struct InequalityData
    q::DeviceVector
    r::DeviceVector
    s::DeviceVector
    t::DeviceVector
    n::Int
end
function InequalityData(ctx::HipContext, xl::Vector{Float64}, xu::Vector{Float64})
    length(xl) == length(xu) || error("xl and xu are of different lengths")
    n = length(xl)
    q, r, s, t = (DeviceVector(ctx, n) for _ in 1:4)
    dxl = upload!(DeviceVector(ctx, n), xl); dxu = upload!(DeviceVector(ctx, n), xu)
    check(ctx, c_ineq_data_build(ctx.h, dxl.h, dxu.h, q.h, r.h, s.h, t.h))
    return InequalityData(q, r, s, t, n)
end
cineq(id::InequalityData) = CIneqData(id.q.h, id.r.h, id.s.h, id.t.h, id.n)

# InequalityDecomp (src/inequality_helper.jl:10-19).  The reference's 2N x M factor U is held as the N x M matrix Z plus
# the row scalings sx = Dy.^2, sy = -Dx.*Dy (U = [sx .* Z; sy .* Z]).
mutable struct InequalityDecomp
    ctx::HipContext
    N::Int
    M::Int
    Z::Union{Nothing,DeviceMatrix}      # nothing: the basis stays in factored form U = [sx; sy] .* (Jct W)
    Σ::Vector{Float64}
    Vt::Matrix{Float64}
    Dx::DeviceVector
    Dy::DeviceVector
    S::DeviceVector
    sx::DeviceVector
    sy::DeviceVector
    Jct::DeviceMatrix
    rank::Int
    W::Union{Nothing,Matrix{Float64}}
    Jsp::Ptr{Cvoid}            # sparse twin of Jct's leading columns (lfpsqp_spmat handle): projcg! on the nonzeros; C_NULL otherwise
end
InequalityDecomp(ctx::HipContext, N::Integer, M::Integer, Jct::DeviceMatrix, Z::Union{Nothing,DeviceMatrix}=DeviceMatrix(ctx, N, M)) =
    InequalityDecomp(ctx, N, M, Z, zeros(M), zeros(M, M), (DeviceVector(ctx, N) for _ in 1:5)..., Jct, M, nothing, C_NULL)
# Q = [[diag Dx; diag Dy], U[:, 1:rank]] (InequalityDecompProject, :25-27, :161-212): projcg!'s U with bounds
struct InequalityDecompProject
    idecomp::InequalityDecomp
end
cbasis(Q::InequalityDecompProject) = (d = Q.idecomp;
    d.W === nothing ? CBasis(d.Z.h, d.rank, d.Dx.h, d.Dy.h, d.sx.h, d.sy.h, C_NULL, C_NULL, C_NULL, C_NULL) :
                      CBasis(zhandle(d.Z), d.rank, d.Dx.h, d.Dy.h, d.sx.h, d.sy.h, d.Jct.h, pointer(d.W), C_NULL, d.Jsp))
const AnyBasis = Union{DeviceBasis,InequalityDecompProject}
ncols(U::DeviceBasis) = U.ncols
ncols(Q::InequalityDecompProject) = Q.idecomp.rank
# [w; t] = Q'v and y = a Q [w; t] + b y  (:161-212); ONE pass over the N x M matrix Z each
q_gemv_t!(w::DeviceVector, t::DeviceVector, Q::InequalityDecompProject, v::DeviceVector) =
    (check(v.ctx, c_q_gemv_t(v.ctx.h, Ref(cbasis(Q)), v.h, w.h, t.h)); nothing)
q_gemv_n!(y::DeviceVector, Q::InequalityDecompProject, w::Union{Nothing,DeviceVector}, t::DeviceVector, a::Real=1.0, b::Real=0.0) =
    (check(y.ctx, c_q_gemv_n(y.ctx.h, Ref(cbasis(Q)), Float64(a), w === nothing ? C_NULL : w.h, t.h, Float64(b), y.h)); y)

generate_initial_y!(xaug::DeviceVector, id::InequalityData) = (check(xaug.ctx, c_generate_initial_y(xaug.ctx.h, xaug.h, Ref(cineq(id)))); xaug)
inequality_gradient!(d::InequalityDecomp, xaug::DeviceVector, id::InequalityData) =
    (check(d.ctx, c_inequality_gradient(d.ctx.h, xaug.h, Ref(cineq(id)), d.Dx.h, d.Dy.h, d.S.h, d.sx.h, d.sy.h)); d)
y_retract!(xnew::DeviceVector, x::DeviceVector, id::InequalityData) = (check(x.ctx, c_y_retract(x.ctx.h, xnew.h, x.h, Ref(cineq(id)))); xnew)

# ksvd! (src/la_helper.jl:8-34, call sites src/optimize.jl:291/293): thin factorisation of diag(sqrt(w2)) Jct; Jct is NOT
# destroyed.  Returns the rank by the reference's rule (Σ_j >= ϵ_rank, :297-302).  W (optional m x m): Z == Jct*W.
# Jsp (optional SparseMatrix with the entries of the leading Jsp.m columns of Jct): the basis-forming products stream the nonzeros.
# Z === nothing (dense Jct, W required): the basis Z = Jct*W is not formed -- the caller keeps it in factored form, DeviceBasis(nothing, rank, (Jct, W))
function ksvd!(Jct::DeviceMatrix, Z::Union{Nothing,DeviceMatrix}, Σ::Vector{Float64}, Vt::Matrix{Float64}; w2::Union{Nothing,DeviceVector}=nothing,
               ϵ_rank::Float64=1e-10, W::Union{Nothing,Matrix{Float64}}=nothing, Jsp=nothing, Vt_prev::Union{Nothing,Matrix{Float64}}=nothing,
               rhs::Union{Nothing,DeviceVector}=nothing, Jte::Union{Nothing,Vector{Float64}}=nothing, G::Union{Nothing,Matrix{Float64}}=nothing)
    # rhs / Jte (dense Jct only): Jte .= Jct' (sqrt.(w2) .* rhs), summed by the Gram pass itself (lfpsqp_factorize_rhs) -- the outer iteration's
    # Jct'd (src/optimize.jl:306) without a GEMV-T pass of its own
    rank = Ref{Int64}(0)
    # warm start of the small eigenproblem from the previous outer iteration's Vt (lfpsqp_factorize_hint; ignored unless orthogonal)
    Vt_prev !== nothing && size(Vt_prev) == size(Vt) && check(Jct.ctx, c_factorize_hint(Jct.ctx.h, Vt_prev, Int64(size(Vt, 1))))
    if Jsp === nothing && rhs !== nothing
        # (G, optional m x m: receives the Gram matrix the factors come from -- U'U = W'GW for the tangent step)
        check(Jct.ctx, c_factorize_rhs(Jct.ctx.h, Jct.h, w2 === nothing ? C_NULL : w2.h, zhandle(Z), Σ, Vt, W === nothing ? C_NULL : W, rank, ϵ_rank, rhs.h, Jte,
                                       G === nothing ? Ptr{Float64}(C_NULL) : pointer(G)))
    elseif Jsp === nothing
        check(Jct.ctx, c_factorize(Jct.ctx.h, Jct.h, w2 === nothing ? C_NULL : w2.h, zhandle(Z), Σ, Vt, W === nothing ? C_NULL : W, rank, ϵ_rank))
    else
        check(Jct.ctx, c_factorize_sp(Jct.ctx.h, Jsp.h, Jct.h, w2 === nothing ? C_NULL : w2.h, zhandle(Z), Σ, Vt, W === nothing ? C_NULL : W, rank, ϵ_rank))
    end
    return Int(rank[])
end

# =====================================================================================================================
# 4. projcg!, retractions, pcg!
# =====================================================================================================================
struct ProjCGWork              # ProjCGWork(n, m), src/projcg.jl:1-11 (three n-vectors suffice on the device; Av: general operators)
    g::DeviceVector
    d::DeviceVector
    rp::DeviceVector
    Utr::DeviceVector
    Av::DeviceVector
end
# (Utr: 3 m + 8 entries -- behind the m coefficients lfpsqp_tangent_step parks the sums of LFPSQP_PROJCG_START_PROJECTED)
ProjCGWork(like::DeviceVector, m::Integer) =
    ProjCGWork(similar_device(like), similar_device(like), similar_device(like), DeviceVector(like.ctx, 3 * max(m, 1) + 8), similar_device(like))
cwork(w::ProjCGWork) = CWork(w.g.h, w.d.h, w.rp.h, w.Utr.h)

# projcg!(x, λ, A, U, b, c; tol, maxit, work) -> (i, nr)  (src/projcg.jl:40-121), fused on the device for a diagonal A ...
function projcg!(x::DeviceVector, λ::Union{Nothing,DeviceVector}, A::DiagOperator, U::AnyBasis, b::DeviceVector, c::Union{Nothing,DeviceVector};
                 tol::Float64=1e-6, maxit::Int=length(b) + ncols(U), work::ProjCGWork=ProjCGWork(x, ncols(U)), n_global::Int=length(b),
                 start_given::Bool=false, start_projected::Bool=false)
    iters = Ref{Int64}(0); nr = Ref{Float64}(0.0)
    flags = (λ === nothing ? Cint(0) : LFPSQP_PROJCG_WANT_LAMBDA) | (start_given ? LFPSQP_PROJCG_START_GIVEN : Cint(0)) |
            (start_projected ? LFPSQP_PROJCG_START_PROJECTED : Cint(0))
    GC.@preserve U begin
        check(x.ctx, c_projcg(x.ctx.h, x.h, λ === nothing ? C_NULL : λ.h, Ref(CDiagOp(A.a0, A.dg === nothing ? C_NULL : A.dg.h)), Ref(cbasis(U)),
                              b.h, c === nothing ? C_NULL : c.h, tol, Int64(maxit), Int64(n_global), flags,
                              Ref(cwork(work)), iters, nr))
    end
    return Int(iters[]), nr[]          # (i, nr) exactly like the reference; nr == Inf and λ .== NaN on negative curvature
end
function projcg!(x::DeviceVector, λ::Union{Nothing,DeviceVector}, A::LowRankOperator, U::AnyBasis, b::DeviceVector, c::Union{Nothing,DeviceVector};
                 tol::Float64=1e-6, maxit::Int=length(b) + ncols(U), work::ProjCGWork=ProjCGWork(x, ncols(U)), n_global::Int=length(b))
    iters = Ref{Int64}(0); nr = Ref{Float64}(0.0)
    GC.@preserve U A begin
        check(x.ctx, c_projcg_lowrank(x.ctx.h, x.h, λ === nothing ? C_NULL : λ.h,
                                      Ref(CLowRankOp(A.a0, A.dg === nothing ? C_NULL : A.dg.h, A.V.h, Int64(A.k), pointer(A.σ))), Ref(cbasis(U)),
                                      b.h, c === nothing ? C_NULL : c.h, tol, Int64(maxit), Int64(n_global), λ === nothing ? Cint(0) : LFPSQP_PROJCG_WANT_LAMBDA,
                                      Ref(cwork(work)), iters, nr))
    end
    return Int(iters[]), nr[]
end
# Tridiagonal Hessian: the one-pass iteration where its shape exists (plain dense basis, 4 .. 1024 columns, one rank); otherwise the callback
# loop below with mul! as the operator -- the same iterates, two passes per iteration.
function projcg!(x::DeviceVector, λ::Union{Nothing,DeviceVector}, A::TridiagonalOperator, U::AnyBasis, b::DeviceVector, c::Union{Nothing,DeviceVector};
                 tol::Float64=1e-6, maxit::Int=length(b) + ncols(U), work::ProjCGWork=ProjCGWork(x, ncols(U)), n_global::Int=length(b),
                 Av::DeviceVector=DeviceVector(x.ctx, length(b)), start_given::Bool=false)
    iters = Ref{Int64}(0); nr = Ref{Float64}(0.0)
    flags = (λ === nothing ? Cint(0) : LFPSQP_PROJCG_WANT_LAMBDA) | (start_given ? LFPSQP_PROJCG_START_GIVEN : Cint(0))
    rc = GC.@preserve U A c_projcg_tridiag(x.ctx.h, x.h, λ === nothing ? C_NULL : λ.h, Ref(ctridiag(A)), Av.h, Ref(cbasis(U)),
                                           b.h, c === nothing ? C_NULL : c.h, tol, Int64(maxit), Int64(n_global), flags, Ref(cwork(work)), iters, nr)
    if rc == LFPSQP_ERR_UNSUPPORTED && !start_given
        return projcg!(x, λ, (dest, src) -> mul!(dest, A, src), U, b, c; tol=tol, maxit=maxit, work=work, n_global=n_global)
    end
    check(x.ctx, rc)
    return Int(iters[]), nr[]
end
# ... and for a GENERAL operator A(dest, src) on device vectors (the LinearMap closure of src/optimize.jl:228-230): the same
# device-resident loop, A called back once per iteration (lfpsqp_projcg_op); its body may queue device work and return.
mutable struct OpBox           # what the C callback needs: the Julia closure and the vectors behind the handles it is given
    apply::Function
    known::Dict{Ptr{Cvoid},DeviceVector}
    err::Any
end
function _op_trampoline(user::Ptr{Cvoid}, src::Ptr{Cvoid}, dest::Ptr{Cvoid})::Cint
    box = unsafe_pointer_to_objref(user)::OpBox
    try
        box.apply(box.known[dest], box.known[src])
        return Cint(0)
    catch e                         # never unwind through C
        box.err = e
        return Cint(1)
    end
end
function projcg!(x::DeviceVector, λ::Union{Nothing,DeviceVector}, A::Function, U::AnyBasis, b::DeviceVector, c::Union{Nothing,DeviceVector};
                 tol::Float64=1e-6, maxit::Int=length(b) + ncols(U), work::ProjCGWork=ProjCGWork(x, ncols(U)), n_global::Int=length(b))
    iters = Ref{Int64}(0); nr = Ref{Float64}(0.0)
    box = OpBox(A, Dict(x.h => x, work.d.h => work.d, work.Av.h => work.Av), nothing)
    cb = @cfunction(_op_trampoline, Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}))
    GC.@preserve box U begin
        rc = c_projcg_op(x.ctx.h, x.h, λ === nothing ? C_NULL : λ.h, cb, pointer_from_objref(box), work.Av.h, Ref(cbasis(U)), b.h,
                         c === nothing ? C_NULL : c.h, tol, Int64(maxit), Int64(n_global), λ === nothing ? Cint(0) : LFPSQP_PROJCG_WANT_LAMBDA,
                         Ref(cwork(work)), iters, nr)
        box.err === nothing || throw(box.err)
        check(x.ctx, rc)
    end
    return Int(iters[]), nr[]
end

# Device-resident equality constraints of the BASELINE configs: c(x) = [J x - b; sum_{i<=n_x} x_i^2 - R2 - x[slack]] (lfpsqp_constraints)
struct DeviceConstraints
    Jct::DeviceMatrix
    m_lin::Int
    b::Vector{Float64}
    has_ball::Bool
    R2::Float64
    n_x::Int
    slack_row::Int              # 0-based LOCAL row of the slack variable, -1 if another rank owns it
    Jsp::Ptr{Cvoid}             # optional SparseMatrix handle with the entries of Jct[:, 1:m_lin] (C_NULL: dense only)
    ew::Any                     # nothing, or the Elementwise description that makes the first m_lin constraints nonlinear (below)
end
DeviceConstraints(Jct, m_lin, b, has_ball, R2, n_x, slack_row) = DeviceConstraints(Jct, m_lin, b, has_ball, R2, n_x, slack_row, C_NULL, nothing)
DeviceConstraints(Jct, m_lin, b, has_ball, R2, n_x, slack_row, Jsp::Ptr{Cvoid}) = DeviceConstraints(Jct, m_lin, b, has_ball, R2, n_x, slack_row, Jsp, nothing)
nconstraints(c::DeviceConstraints) = c.m_lin + (c.has_ball ? 1 : 0)
ccons(c::DeviceConstraints) = CConstraints(c.Jct.h, c.m_lin, pointer(c.b), c.has_ball ? 1 : 0, c.R2, c.n_x, c.slack_row, c.Jsp,
                                           c.ew === nothing ? Ptr{CElementwise}(C_NULL) : Base.unsafe_convert(Ptr{CElementwise}, c.ew.cref))

# ---- the device-resident NONLINEAR constraint class (lfpsqp_elementwise): c(x) = A' phi(x) + qw * sum_{i<=n_x} x_i^2 - b -------------
# phi elementwise per variable: kind 0: t, 1: sin t, 2: t^2.  Jct(x) = Diagonal(phi'(x)) A + 2 x qw' is refreshed in place by jac!, the
# constraint part of the Lagrangian Hessian is diagonal (hess_diag!), c! streams A (or its nonzeros) once -- the contract of the reference's
# c! / jac! / hess_lag_vec! (src/autodiff_generators.jl:72-107) without AD.  Covers the reference's own nonlinear test systems
# (test/test_retractions.jl:1-54): sin_system_constraints, sphere_system_constraints.
mutable struct Elementwise
    A::Union{Nothing,DeviceMatrix}
    Asp::Union{Nothing,SparseMatrix}
    Jsp::Union{Nothing,SparseMatrix}       # clone of Asp holding the current gradients
    kind::Union{Nothing,DeviceVector}
    qw::Union{Nothing,Vector{Float64}}
    work::Union{Nothing,DeviceVector}
    cref::Base.RefValue{CElementwise}
end
# stream (dense A with a kind and / or qw; default: whenever the one-pass kernels cover the shape): Jct is a VIEW Diagonal(phi'(x)) A + 2 x qw' of A -- jac!
# refreshes the n-vectors phi'(x) and 2 x instead of rewriting an n x m matrix, every solver streams the constant A
function ElementwiseConstraints(ctx::HipContext, A::Union{DeviceMatrix,SparseMatrix}, b::Vector{Float64};
                                kind::Union{Nothing,Vector{Float64}}=nothing, qw::Union{Nothing,Vector{Float64}}=nothing,
                                stream::Union{Nothing,Bool}=nothing)
    n, m = A.n, A.m
    sparse = A isa SparseMatrix
    can_stream = !sparse && (kind !== nothing || qw !== nothing) && m >= 1
    stream === true && !can_stream && throw(ArgumentError("streamed gradients need a dense A with a kind or a quadratic term"))
    streamed = stream === nothing ? (can_stream && factored_basis_supported(ctx, A)) : stream
    Jct = streamed ? matrix_view(A; rs=(kind === nothing ? nothing : fill!(DeviceVector(ctx, n), 1.0)), u=(qw === nothing ? nothing : DeviceVector(ctx, n)),
                                 w=(qw === nothing ? nothing : upload!(DeviceVector(ctx, m), qw))) : DeviceMatrix(ctx, n, m)
    Jsp = sparse ? clone(A) : nothing
    kd = kind === nothing ? nothing : upload!(DeviceVector(ctx, n), kind)
    work = sparse ? DeviceVector(ctx, n) : nothing
    qwc = qw === nothing ? nothing : copy(qw)
    cref = Ref(CElementwise(sparse ? C_NULL : A.h, sparse ? A.h : C_NULL, kd === nothing ? C_NULL : kd.h,
                            qwc === nothing ? Ptr{Float64}(C_NULL) : pointer(qwc), work === nothing ? C_NULL : work.h))
    ew = Elementwise(sparse ? nothing : A, sparse ? A : nothing, Jsp, kd, qwc, work, cref)
    return DeviceConstraints(Jct, m, copy(b), false, 0.0, n, -1, sparse ? Jsp.h : C_NULL, ew)
end
# generate_sin_system(n, m) (test/test_retractions.jl:34-54): c_i = x[2i] - sin(x[2i-1])
function sin_system_constraints(ctx::HipContext, n::Int, m::Int)
    I = vcat(2 .* (1:m), 2 .* (1:m) .- 1); J = vcat(1:m, 1:m); V = vcat(ones(m), -ones(m))
    kind = zeros(n); kind[1:2:2m] .= 1.0
    return ElementwiseConstraints(ctx, SparseMatrix(ctx, n, m, collect(I), collect(J), V), zeros(m); kind=kind)
end
# generate_sphere_system (test/test_retractions.jl:1-31): c_i = |x - center_i|^2 - R_i^2 = x'x - 2 center_i'x + |center_i|^2 - R_i^2
function sphere_system_constraints(ctx::HipContext, centers::Matrix{Float64}, Rs::Vector{Float64})
    n, m = size(centers)
    A = upload!(DeviceMatrix(ctx, n, m), -2.0 .* centers)
    return ElementwiseConstraints(ctx, A, Rs .^ 2 .- vec(sum(abs2, centers; dims=1)); qw=ones(m))
end
# hx .+= diag of sum_j lam_j grad^2 c_j(x): phi''(x) .* (A lam) + 2 qw'lam (+ 2 lam_ball) on the user's variables
function hess_diag!(c::DeviceConstraints, hx::DeviceVector, x::DeviceVector, λ::Vector{Float64})
    GC.@preserve c check(x.ctx, c_constraints_hess_diag(x.ctx.h, Ref(ccons(c)), x.h, λ, hx.h))
    return hx
end
function (c::DeviceConstraints)(cval::Vector{Float64}, x::DeviceVector)                       # c!(cval, x)
    GC.@preserve c check(x.ctx, c_constraints_eval(x.ctx.h, Ref(ccons(c)), x.h, cval))
    return cval
end
function jac!(c::DeviceConstraints, Jct::DeviceMatrix, cval::Vector{Float64}, x::DeviceVector)    # jac!(Jct, cval, x)
    GC.@preserve c check(x.ctx, c_constraints_jac(x.ctx.h, Ref(ccons(c)), x.h, Jct.h, cval))
    return cval
end
# the gradients only (cval == NULL): for a caller that holds c(x) already -- the accepted retraction of the line search returned it
function jac!(c::DeviceConstraints, Jct::DeviceMatrix, ::Nothing, x::DeviceVector)
    GC.@preserve c check(x.ctx, c_constraints_jac(x.ctx.h, Ref(ccons(c)), x.h, Jct.h, Ptr{Float64}(C_NULL)))
    return nothing
end

# A host c!(cval, x::Vector) behind the C callback of lfpsqp_retract_nr / lfpsqp_retract_pp: x is downloaded per evaluation
mutable struct CfunBox
    ctx::HipContext
    c!::Function
    nrows::Int
    m::Int
    err::Any
end
function _cfun_trampoline(user::Ptr{Cvoid}, xvec::Ptr{Cvoid}, cval::Ptr{Float64})::Cint
    box = unsafe_pointer_to_objref(user)::CfunBox
    try
        xh = Vector{Float64}(undef, box.nrows)
        c_vec_download(box.ctx.h, xvec, Int64(0), xh, Int64(box.nrows)) == 0 || error("download of the iterate failed")
        box.c!(unsafe_wrap(Array, cval, box.m), xh)
        return Cint(0)
    catch e
        box.err = e
        return Cint(1)
    end
end

abstract type RetractionMethod end
struct Euclidean <: RetractionMethod end                                      # src/retractions.jl:51-52
struct YRetract <: RetractionMethod                                           # :54-56
    idata::InequalityData
end
mutable struct NR <: RetractionMethod                                         # :10-19
    U::Union{Nothing,AnyBasis}
    Σ::Vector{Float64}
    Vt::Matrix{Float64}
    tol::Float64
    maxiter::Int
    ineq::Bool
    idata::Union{Nothing,InequalityData}
end
struct ProjPenaltyWork                                                        # :21-33 (J itself is the shared device Jct)
    r::DeviceVector
    p::DeviceVector
    z::DeviceVector
    dx::DeviceVector
    g::DeviceVector
    tmp_m::DeviceVector
    tmp_w::Union{Nothing,DeviceVector}
    h::Union{Nothing,DeviceVector}
    DxS::Union{Nothing,DeviceVector}
    DyS::Union{Nothing,DeviceVector}
    ones::Union{Nothing,DeviceVector}
    zeros::Union{Nothing,DeviceVector}
    q::Union{Nothing,DeviceVector}          # DeviceOptions.pp_precondition: the inner solves run lfpsqp_pcg_pre
    i11::Union{Nothing,DeviceVector}
    i12::Union{Nothing,DeviceVector}
    i22::Union{Nothing,DeviceVector}
end
# `against`: the constraint-gradient matrix the pcg! iteration streams -- the five n-vectors then come from one allocation chosen by the
# library's placement policy against it (lfpsqp_vecs_alloc_placed), as ProjCGWork's do
function ProjPenaltyWork(like::DeviceVector, m::Integer, N::Integer, ineq::Bool; against::Union{Nothing,DeviceMatrix}=nothing)
    ctx = like.ctx
    v5 = (against !== nothing && ctx.options.placement_tries > 1) ? vectors_placed(ctx, against, N, 5; N=(ineq ? N : 0)) :
         [similar_device(like) for _ in 1:5]
    pre = ctx.options.pp_precondition
    q = pre ? similar_device(like) : nothing
    ineq || return ProjPenaltyWork(v5..., DeviceVector(ctx, max(m, 1)), nothing, nothing, nothing, nothing, nothing, nothing, q, nothing, nothing, nothing)
    return ProjPenaltyWork(v5..., DeviceVector(ctx, max(m, 1)), DeviceVector(ctx, N), DeviceVector(ctx, N), DeviceVector(ctx, N), DeviceVector(ctx, N),
                           fill!(DeviceVector(ctx, N), 1.0), DeviceVector(ctx, N), q, pre ? DeviceVector(ctx, N) : nothing,
                           pre ? DeviceVector(ctx, N) : nothing, pre ? DeviceVector(ctx, N) : nothing)
end
_h(v) = v === nothing ? C_NULL : v.h
cppwork(w::ProjPenaltyWork) = CPPWork(w.r.h, w.p.h, w.z.h, w.dx.h, w.g.h, w.tmp_m.h, _h(w.tmp_w), _h(w.h), _h(w.DxS), _h(w.DyS), _h(w.ones), _h(w.zeros),
                                      _h(w.q), _h(w.i11), _h(w.i12), _h(w.i22), Cint(w.q === nothing ? 0 : 1))
mutable struct ProjPenalty <: RetractionMethod                                # :35-49
    jac!::Any                  # DeviceConstraints (device-resident jac!) or a host jac!(J, cval, x)
    m::Int
    rank::Int
    μ0::Float64
    tol::Float64
    maxiter::Int
    maxiter_pcg::Int
    work::ProjPenaltyWork
    ineq::Bool
    idecomp::InequalityDecomp
    idata::Union{Nothing,InequalityData}
end

# retract!(cval, xnew, c!, xtilde, x, method) -> (flag, iter1, iter2)
function retract!(cval::Vector{Float64}, xnew::DeviceVector, c!, xtilde::DeviceVector, x::DeviceVector, ::Euclidean)      # :61-65
    copyto!(xnew, xtilde)
    return 0, 0, 0
end
function retract!(cval::Vector{Float64}, xnew::DeviceVector, c!, xtilde::DeviceVector, x::DeviceVector, method::YRetract)   # :67-72
    copyto!(xnew, xtilde)
    y_retract!(xnew, x, method.idata)
    return 0, 0, 0
end
function retract!(cval::Vector{Float64}, xnew::DeviceVector, c!, xtilde::DeviceVector, x::DeviceVector, method::NR)         # :75-177
    flag = Ref{Cint}(0); iters = Ref{Int64}(0)
    m = length(method.Σ)
    idref = method.ineq ? Ref(cineq(method.idata)) : nothing
    idp = method.ineq ? Base.unsafe_convert(Ptr{CIneqData}, idref) : Ptr{CIneqData}(C_NULL)
    if c! isa DeviceConstraints
        consref = Ref(ccons(c!))
        GC.@preserve method c! consref idref begin
            check(x.ctx, c_retract_nr(x.ctx.h, Ref(cbasis(method.U)), method.Σ, method.Vt, Int64(m), Base.unsafe_convert(Ptr{CConstraints}, consref),
                                      C_NULL, C_NULL, idp, xtilde.h, x.h, xnew.h, method.tol, Int64(method.maxiter), cval, flag, iters))
        end
    else
        box = CfunBox(x.ctx, c!, method.ineq ? method.idata.n : x.n, m, nothing)
        cb = @cfunction(_cfun_trampoline, Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}))
        GC.@preserve method box idref begin
            rc = c_retract_nr(x.ctx.h, Ref(cbasis(method.U)), method.Σ, method.Vt, Int64(m), Ptr{CConstraints}(C_NULL), cb, pointer_from_objref(box), idp,
                              xtilde.h, x.h, xnew.h, method.tol, Int64(method.maxiter), cval, flag, iters)
            box.err === nothing || throw(box.err)
            check(x.ctx, rc)
        end
    end
    return Int(flag[]), Int(iters[]), 0
end
# trial retractions per pass for a search: DeviceOptions.ls_batch, never more than the library takes for this shape
# (lfpsqp_retract_nr_batch_width: 4 in the exact mode; 16 / 8 on the matrix cores; 0 = cannot batch)
function batch_width(ctx::HipContext, retract_method, c!, prev_failures::Int)
    (retract_method isa NR && c! isa DeviceConstraints) || return 1
    opt = ctx.options.ls_batch
    opt == 1 && return 1
    w = Ref{Cint}(0)
    GC.@preserve retract_method c! check(ctx, c_retract_nr_batch_width(ctx.h, Ref(cbasis(retract_method.U)), Ref(ccons(c!)), w))
    width = Int(w[])
    width < 2 && return 1
    return opt > 1 ? min(opt, width) : min(width, max(4, prev_failures + 2))
end
# the widest pass of a search: a pass whose trials ALL failed is followed by one twice as wide, up to this
function batch_cap(ctx::HipContext, retract_method, c!)
    (retract_method isa NR && c! isa DeviceConstraints) || return 1
    opt = ctx.options.ls_batch
    opt == 1 && return 1
    w = Ref{Cint}(0)
    GC.@preserve retract_method c! check(ctx, c_retract_nr_batch_width(ctx.h, Ref(cbasis(retract_method.U)), Ref(ccons(c!)), w))
    width = Int(w[])
    return width < 2 ? 1 : (opt > 1 ? min(opt, width) : width)
end
# several trial points of one Armijo search retracted together (lfpsqp_retract_nr_batch); `nothing` when this configuration cannot
function retract_nr_batch!(cvals::Matrix{Float64}, xnews::Vector{DeviceVector}, c!::DeviceConstraints, xtildes::Vector{DeviceVector}, x::DeviceVector, method::NR)
    nb = length(xnews)
    flags = zeros(Cint, nb); iters = zeros(Int64, nb)
    xt = [v.h for v in xtildes]; xn = [v.h for v in xnews]
    idref = method.ineq ? Ref(cineq(method.idata)) : nothing
    idp = method.ineq ? Base.unsafe_convert(Ptr{CIneqData}, idref) : Ptr{CIneqData}(C_NULL)
    rc = Cint(0)
    GC.@preserve method c! xt xn idref begin
        rc = c_retract_nr_batch(x.ctx.h, Ref(cbasis(method.U)), method.Σ, method.Vt, Int64(length(method.Σ)), Ref(ccons(c!)), idp, Cint(nb), xt, x.h, xn,
                                method.tol, Int64(method.maxiter), cvals, flags, iters)            # cvals is m x nb (column b = cval of trial b)
    end
    rc == LFPSQP_ERR_UNSUPPORTED && return nothing
    check(x.ctx, rc)
    return [(Int(flags[b]), Int(iters[b]), 0) for b in 1:nb]
end
# the reference's DEFAULT retraction (:265-441), ONE C call; c! / jac! device-resident or host callables
mutable struct JacBox
    ctx::HipContext
    c!::Any
    jac!::Any
    Jct::DeviceMatrix
    nrows::Int
    m::Int
    err::Any
end
function _pp_c_trampoline(user::Ptr{Cvoid}, xvec::Ptr{Cvoid}, cval::Ptr{Float64})::Cint
    box = unsafe_pointer_to_objref(user)::JacBox
    try
        xh = Vector{Float64}(undef, box.nrows)
        c_vec_download(box.ctx.h, xvec, Int64(0), xh, Int64(box.nrows)) == 0 || error("download of the iterate failed")
        box.c!(unsafe_wrap(Array, cval, box.m), xh)
        return Cint(0)
    catch e
        box.err = e
        return Cint(1)
    end
end
function _pp_jac_trampoline(user::Ptr{Cvoid}, xvec::Ptr{Cvoid}, Jct::Ptr{Cvoid}, cval::Ptr{Float64})::Cint
    box = unsafe_pointer_to_objref(user)::JacBox
    try
        xh = Vector{Float64}(undef, box.nrows)
        c_vec_download(box.ctx.h, xvec, Int64(0), xh, Int64(box.nrows)) == 0 || error("download of the iterate failed")
        J = zeros(box.m, box.nrows)
        box.jac!(J, unsafe_wrap(Array, cval, box.m), xh)
        upload!(box.Jct, Matrix(J'))                   # the device keeps only Jct = J'
        return Cint(0)
    catch e
        box.err = e
        return Cint(1)
    end
end
function retract!(cval::Vector{Float64}, xnew::DeviceVector, c!, xtilde::DeviceVector, x::DeviceVector, pp::ProjPenalty)
    flag = Ref{Cint}(0); iters = Ref{Int64}(0); pcg_iters = Ref{Int64}(0)
    d = pp.idecomp
    idref = pp.ineq ? Ref(cineq(pp.idata)) : nothing
    idp = pp.ineq ? Base.unsafe_convert(Ptr{CIneqData}, idref) : Ptr{CIneqData}(C_NULL)
    Dx, Dy, S = pp.ineq ? (d.Dx.h, d.Dy.h, d.S.h) : (C_NULL, C_NULL, C_NULL)
    if c! isa DeviceConstraints
        consref = Ref(ccons(c!))
        GC.@preserve pp c! consref idref begin
            check(x.ctx, c_retract_pp(x.ctx.h, Base.unsafe_convert(Ptr{CConstraints}, consref), C_NULL, C_NULL, C_NULL, d.Jct.h, Int64(pp.m), idp, Dx, Dy, S,
                                      xtilde.h, x.h, xnew.h, pp.μ0, pp.tol, Int64(pp.maxiter), Int64(pp.maxiter_pcg), Ref(cppwork(pp.work)), cval,
                                      flag, iters, pcg_iters))
        end
    else
        box = JacBox(x.ctx, c!, pp.jac!, d.Jct, d.N, pp.m, nothing)
        cbc = @cfunction(_pp_c_trampoline, Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}))
        cbj = @cfunction(_pp_jac_trampoline, Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}))
        GC.@preserve pp box idref begin
            rc = c_retract_pp(x.ctx.h, Ptr{CConstraints}(C_NULL), cbc, cbj, pointer_from_objref(box), d.Jct.h, Int64(pp.m), idp, Dx, Dy, S,
                              xtilde.h, x.h, xnew.h, pp.μ0, pp.tol, Int64(pp.maxiter), Int64(pp.maxiter_pcg), Ref(cppwork(pp.work)), cval,
                              flag, iters, pcg_iters)
            box.err === nothing || throw(box.err)
            check(x.ctx, rc)
        end
    end
    return Int(flag[]), Int(iters[]), Int(pcg_iters[])
end
# pcg!(μ, J, no_precondition, x, r, p, z, tmp_m, tol, maxiter) (:179-246) with J' given in the lfpsqp_basis form
function pcg!(μ::Float64, Jt::CBasis, x::DeviceVector, r::DeviceVector, p::DeviceVector, z::DeviceVector, tmp_w::Union{Nothing,DeviceVector},
              tmp_m::DeviceVector, tol::Float64, maxiter::Int)
    flag = Ref{Cint}(0); iters = Ref{Int64}(0)
    check(x.ctx, c_pcg(x.ctx.h, μ, Ref(Jt), x.h, r.h, p.h, z.h, _h(tmp_w), tmp_m.h, tol, Int64(maxiter), flag, iters))
    return Int(flag[]), Int(iters[])
end

# pcg! with M! = proj_precondition!(z, r, μ, U, Σ, rank, tmp_m) (:248-257; the call commented out at :374), fused on the device: U = Jct W is
# known through its generator, the whole solve is one lfpsqp_pcg_pre call with K = μ W diag(σ² / (μ + σ²)) W'.  q: work n-vector.
struct ProjPrecondition
    W::Matrix{Float64}
    Σ::Vector{Float64}
    rank::Int
    q::DeviceVector
end
function pcg!(μ::Float64, Jt::CBasis, M::ProjPrecondition, x::DeviceVector, r::DeviceVector, p::DeviceVector, z::DeviceVector, tol::Float64, maxiter::Int)
    Wr = M.W[:, 1:M.rank]
    s2 = M.Σ[1:M.rank] .^ 2
    K = μ .* (Wr .* (s2 ./ (μ .+ s2))') * Wr'
    flag = Ref{Cint}(0); iters = Ref{Int64}(0)
    GC.@preserve K begin
        pc = CPcgPrecond(pointer(K), C_NULL, C_NULL, C_NULL, M.q.h)
        check(x.ctx, c_pcg_pre(x.ctx.h, μ, Ref(Jt), Ref(pc), x.h, r.h, p.h, z.h, tol, Int64(maxiter), flag, iters))
    end
    return Int(flag[]), Int(iters[])
end

# =====================================================================================================================
# 5. Parameters and linesearch (src/LFPSQP.jl:27-81, src/linesearch.jl) -- host control flow, device vectors
# =====================================================================================================================
@enum DisplayOption off iter
@enum LinesearchOption armijo exact
@enum TerminationCondition f_tol x_tol kkt_tol max_iter armijo_error
struct TerminationInfo
    condition::TerminationCondition
    f_diff::Float64
    step_diff::Float64
    kkt_diff::Float64
    iter::Int
end
Base.@kwdef mutable struct LFPSQPParams
    α::Float64 = 1.0
    β::Float64 = 0.0
    t_β::Int = 0
    s::Float64 = 0.5
    σ::Float64 = 1e-4
    ϵ_c::Float64 = 1e-6
    ϵ_f::Float64 = 1e-6
    ϵ_x::Float64 = 0.0
    ϵ_kkt::Float64 = 1e-6
    ϵ_rank::Float64 = 1e-10
    maxiter::Int = 10000
    maxiter_retract::Int = 100
    maxiter_pcg::Int = 100
    μ0::Float64 = 1e-2
    disable_linesearch::Bool = false
    do_project_retract::Bool = true
    disp::DisplayOption = iter
    callback::Union{Nothing,Function} = nothing
    callback_period::Int = 100
    linesearch::LinesearchOption = armijo
    do_newton::Bool = true
    tn_maxiter::Int = 10000
    tn_κ::Float64 = 0.5
end

mutable struct ArmijoWork       # src/linesearch.jl:1-5
    xtilde::DeviceVector
    xts::Vector{DeviceVector}
    xns::Vector{DeviceVector}
    prev_failed::Bool
    prev_failures::Int
end
ArmijoWork(like::DeviceVector) = ArmijoWork(similar_device(like), DeviceVector[], DeviceVector[], false, 0)
struct ExactLinesearchWork      # :7-14
    tmp_n1::DeviceVector
    tmp_n2::DeviceVector
    tmp_n3::DeviceVector
    tmp_n4::DeviceVector
    xts::Vector{DeviceVector}   # look-ahead trial points of the shrinking phase (allocated on first use)
    xns::Vector{DeviceVector}
end
ExactLinesearchWork(like::DeviceVector) = ExactLinesearchWork((similar_device(like) for _ in 1:4)..., DeviceVector[], DeviceVector[])

# armijo!(xnew, x, n, d, g, f, fval, retract_method, cval, c!, param, work) (src/linesearch.jl:32-89).  After the first failed
# retraction of a search (or from its first trial when the previous search had failures) the next `ls_batch` trial steps of the
# reference's own sequence α s, α s², ... are retracted together and consumed in the reference's order: same accepted step,
# same counts.
function armijo!(xnew::DeviceVector, x::DeviceVector, n::Int, d::DeviceVector, g::DeviceVector, f, fval::Float64, retract_method, cval::Vector{Float64}, c!,
                 param::LFPSQPParams, work::ArmijoWork)
    f_diff = Inf; step_diff = Inf
    α = param.α
    flag = 0; tot_iter1 = 0; tot_iter2 = 0; newf = 0.0
    ar_dot = dot(d, g)
    xtilde = work.xtilde
    step = xtilde
    ahead = Dict{Float64,Tuple{Int,Int,Int,DeviceVector,Vector{Float64}}}()
    nbatch = batch_width(x.ctx, retract_method, c!, work.prev_failures)
    nbatch_cap = nbatch > 1 ? batch_cap(x.ctx, retract_method, c!) : 1
    failed_once = work.prev_failed
    any_failed = false
    n_failed = 0
    while step_diff > param.ϵ_x
        iter1 = 0; iter2 = 0
        if haskey(ahead, α)
            flag, iter1, iter2, xb, cb = pop!(ahead, α)
            copyto!(xnew, xb); cval .= cb
        else
            got = nothing
            if failed_once && nbatch > 1 && !param.disable_linesearch
                while length(work.xts) < nbatch
                    push!(work.xts, similar_device(x)); push!(work.xns, similar_device(x))
                end
                αs = [α * param.s^(k - 1) for k in 1:nbatch]
                for k in 2:nbatch                       # the reference's own products α*s, (α*s)*s, ...
                    αs[k] = αs[k-1] * param.s
                end
                for k in 1:nbatch
                    waxpby!(work.xts[k], 1.0, x, αs[k], d)
                end
                cvs = zeros(length(cval), nbatch)
                got = retract_nr_batch!(cvs, work.xns[1:nbatch], c!, work.xts[1:nbatch], x, retract_method)
                if got !== nothing
                    for k in 2:nbatch
                        ahead[αs[k]] = (got[k][1], got[k][2], got[k][3], work.xns[k], cvs[:, k])
                    end
                    flag, iter1, iter2 = got[1]
                    copyto!(xnew, work.xns[1]); cval .= cvs[:, 1]
                    all(r -> r[1] > 0, got) && (nbatch = min(nbatch_cap, 2 * nbatch))   # every trial of the pass failed: the next takes twice as many
                else
                    nbatch = 1
                end
            end
            if got === nothing
                waxpby!(xtilde, 1.0, x, α, d)                                       # xtilde = x + α d
                flag, iter1, iter2 = retract!(cval, xnew, c!, xtilde, x, retract_method)
            end
        end
        tot_iter1 += iter1; tot_iter2 += iter2
        if flag > 0                                                               # :57-60
            failed_once = true; any_failed = true; n_failed += 1
            α *= param.s
            continue
        end
        waxpby!(step, 1.0, xnew, -1.0, x)                                           # step = xnew - x
        newf = f(xnew)
        step_diff = norm_head(step, n)                                              # :66
        f_diff = abs(newf - fval)
        param.disable_linesearch && break
        (newf - fval) <= param.σ * α * ar_dot && break                              # :75
        α *= param.s
        if α < 1e-100                                                             # :82
            flag = 99
            break
        end
    end
    work.prev_failed = any_failed
    work.prev_failures = n_failed
    return flag, tot_iter1, tot_iter2, newf, f_diff, step_diff, α
end

# exact_linesearch!(xnew, x, n, d, f, fval, retract_method, cval, c!, param, work) (src/linesearch.jl:107-339): bracketing,
# shrinking and golden-section phases with the reference's rotation of its four work vectors
function exact_linesearch!(xnew::DeviceVector, x::DeviceVector, n::Int, d::DeviceVector, f, fval::Float64, retract_method, cval::Vector{Float64}, c!,
                           param::LFPSQPParams, work::ExactLinesearchWork)
    φ1 = (3 - sqrt(5)) / 2; φ2 = (sqrt(5) - 1) / 2; φ3 = (sqrt(5) + 1) / 2
    Δ = param.α
    flag = 0; tot1 = 0; tot2 = 0
    f_a = 0.0; f_b = 0.0; f_c = 0.0; f_d = 0.0
    a_a = 0.0; a_b = 0.0; a_c = 0.0; a_d = 0.0
    x_a, x_b, x_c, x_d = work.tmp_n1, work.tmp_n2, work.tmp_n3, work.tmp_n4
    step = work.tmp_n1
    do_shrinking = true
    retract_pt! = function (pt::DeviceVector)
        fl, i1, i2 = retract!(cval, xnew, c!, pt, x, retract_method)
        tot1 += i1; tot2 += i2
        copyto!(pt, xnew)
        return fl
    end
    # Shrinking phase (:176-208): the trial steps a_c φ1, a_c φ1², ... are a fixed sequence from the same x, and where it runs most of
    # them fail after the full iteration limit: the next `ls_batch` are retracted together and consumed in the reference's order.
    nbatch = batch_width(x.ctx, retract_method, c!, 2)
    ahead = Dict{Float64,Tuple{Int,Int,Int,DeviceVector,Vector{Float64}}}()
    retract_shrink! = function (pt::DeviceVector, a_next::Float64)
        if haskey(ahead, a_next)
            fl, i1, i2, xb, cb = pop!(ahead, a_next)
            copyto!(xnew, xb); cval .= cb; copyto!(pt, xnew)
            tot1 += i1; tot2 += i2
            return fl
        end
        if nbatch > 1
            while length(work.xts) < nbatch
                push!(work.xts, similar_device(x)); push!(work.xns, similar_device(x))
            end
            αs = [a_next]
            for k in 2:nbatch
                push!(αs, αs[end] * φ1)                       # the reference's own products α_c *= ϕ1
            end
            copyto!(work.xts[1], pt)
            for k in 2:nbatch
                waxpby!(work.xts[k], 1.0, x, αs[k], d)
            end
            cvs = zeros(length(cval), nbatch)
            got = retract_nr_batch!(cvs, work.xns[1:nbatch], c!, work.xts[1:nbatch], x, retract_method)
            if got !== nothing
                for k in 2:nbatch
                    ahead[αs[k]] = (got[k][1], got[k][2], got[k][3], work.xns[k], cvs[:, k])
                end
                fl, i1, i2 = got[1]
                copyto!(xnew, work.xns[1]); cval .= cvs[:, 1]; copyto!(pt, xnew)
                tot1 += i1; tot2 += i2
                return fl
            end
            nbatch = 1
        end
        return retract_pt!(pt)
    end
    copyto!(x_d, x); f_d = fval
    while true
        x_b, x_c, x_d = x_c, x_d, x_b
        f_b, f_c = f_c, f_d
        a_b, a_c = a_c, a_d
        waxpby!(x_d, 1.0, x, a_d + Δ, d)
        flag = retract_pt!(x_d)
        a_d += Δ
        if flag > 0 || a_d > 1.0
            f_d = Inf
            break
        end
        f_d = f(x_d)
        f_d > f_c && break
        do_shrinking = false
        Δ *= φ3
    end
    if do_shrinking
        f_b = fval; a_b = 0.0
        copyto!(x_b, x)
        f_c = Inf; a_c = Δ
        x_d, x_c = x_c, x_d
        while true
            x_d, x_c = x_c, x_d
            f_d = f_c; a_d = a_c
            waxpby!(x_c, 1.0, x, φ1 * a_c, d)
            flag = retract_shrink!(x_c, φ1 * a_c)
            a_c *= φ1
            f_c = (flag > 0 || a_c > 1.0) ? Inf : f(x_c)
            (f_c <= fval || a_c < 1e-100) && break
        end
    end
    f_a, f_b = f_b, f_c
    a_a, a_b = a_b, a_c
    x_a, x_b, x_c = x_b, x_c, x_a
    a_c = a_a + φ2 * (a_d - a_a)
    waxpby!(x_c, 1.0, x, a_c, d)
    flag = retract_pt!(x_c)
    f_c = (flag > 0 || a_c > 1.0) ? Inf : f(x_c)
    nd = norm(d)
    while (a_c - a_b) > 1e-6 * nd
        if f_b < f_c || isinf(f_c)
            x_d, x_c, x_b = x_c, x_b, x_d
            f_d, f_c = f_c, f_b
            a_d, a_c = a_c, a_b
            a_b = a_a + φ1 * (a_d - a_a)
            waxpby!(x_b, 1.0, x, a_b, d)
            flag = retract_pt!(x_b)
            f_b = f(x_b)
        else
            x_a, x_b, x_c = x_b, x_c, x_a
            f_a, f_b = f_b, f_c
            a_a, a_b = a_b, a_c
            a_c = a_a + φ2 * (a_d - a_a)
            waxpby!(x_c, 1.0, x, a_c, d)
            flag = retract_pt!(x_c)
            f_c = (flag > 0 || a_c > 1.0) ? Inf : f(x_c)
        end
    end
    newf = 0.0; α = 0.0
    if f_b < f_c
        copyto!(xnew, x_b); newf = f_b; α = a_b
    else
        copyto!(xnew, x_c); newf = f_c; α = a_c
    end
    waxpby!(step, 1.0, xnew, -1.0, x)
    step_diff = norm_head(step, n)
    f_diff = abs(newf - fval)
    return flag, tot1, tot2, newf, f_diff, step_diff, α
end

# =====================================================================================================================
# 6. optimize (src/optimize.jl:13-443) on device-resident state
# =====================================================================================================================
# Callback contract of the core method (the device analogue of the reference's explicit-derivative method, :119):
#   f(x)                       -> Float64; x is the device iterate (stacked [x; y] with bounds; f looks at the first n entries)
#   grad!(g, x)                writes the first n entries of the device vector g
#   c!                         a DeviceConstraints, or a host c!(cval, x::Vector)
#   jac!(Jct, cval, x)         refreshes the device n x m matrix Jct (= Jc' of the reference) and cval
#   hess_lag_vec!              an object with hess_diag!(hx, x, λ) (diagonal Lagrangian Hessian: fused projcg path), or a
#                              function hess_lag_vec!(dest, src, x, λ_dev) on device vectors (general path, lfpsqp_projcg_op)
function print_iter_header()
    println("   step |          f     ||c||      |Δf|    ||Δx||  |   S iter      res  |   M   iter  (pcg)  |        α  flag")
    println("-"^110)
end
print_first_line(fval, normc) = @printf("      0 | %10.3e  %8.1e                      |                    |                    |               \n", fval, normc)
print_iter(i, fval, normc, fstep, normx, steptype, tn_iter, tn_res, mtype, iter1, iter2, α, flag) =
    @printf("%7d | %10.3e  %8.1e  %8.1e  %8.1e  |  %s %4d %8.1e  |  %s %6d %6d  | %8.1e  %4d\n", i, fval, normc, fstep, normx,
            steptype == 0 ? "GD" : "TN", tn_iter, tn_res, mtype == 0 ? "NR" : "PP", iter1, iter2, α, flag)

function hess_diag! end         # hess_diag!(problem, hx, x, λ): the diagonal of the Lagrangian Hessian, written into the device vector hx
has_hess_diag(h) = hasmethod(hess_diag!, Tuple{typeof(h),DeviceVector,DeviceVector,Vector{Float64}})
# The SPLIT of the diagonal Lagrangian Hessian a device-resident problem class offers: hess_diag! == hess_diag_objective! (no multipliers) followed by
# hess_diag!(hess_constraints(P), hx, x, λ).  optimize_core folds the second half into its tangent-step pass (lfpsqp_tangent_step).
function hess_diag_objective! end
function hess_constraints end
has_hess_split(h) = hasmethod(hess_diag_objective!, Tuple{typeof(h),DeviceVector,DeviceVector}) && hasmethod(hess_constraints, Tuple{typeof(h)})
scalar_hessian(h) = false      # true: grad^2 of the Lagrangian is a multiple of I (projected CG ends after one iteration: no allocation by trial)
# A TRIDIAGONAL Lagrangian Hessian: hess_diag! fills the diagonal, hess_offdiag(problem) returns the device vector of the couplings (entry i couples
# variables i and i+1) or nothing.  The truncated-Newton solves then run projcg! with a TridiagonalOperator: one pass per iteration.
hess_offdiag(h) = nothing

function optimize_core(ctx::HipContext, f, grad!, c!, jac!, hess_lag_vec!, x0::Vector{Float64}, xl, xu, m::Int, param::LFPSQPParams=LFPSQPParams();
                       n_global::Int=length(x0))
    n = length(x0)
    if xl !== nothing && xu !== nothing
        length(xl) == length(xu) == n || error("xl, xu, and x0 must all be the same length")
    end
    ineq = !((xl === nothing && xu === nothing) || (all(xl .== -Inf) && all(xu .== Inf)))       # :146-170
    idata = nothing
    if ineq
        any(xl .> xu) && error("Infeasible: lower bounds cannot be greater than upper bounds")
        idata = InequalityData(ctx, Vector{Float64}(xl), Vector{Float64}(xu))
    end
    newvec() = ineq ? StackedVector(ctx, n) : DeviceVector(ctx, n)
    jsp = (c! isa DeviceConstraints && c!.Jsp != C_NULL) ? (h = c!.Jsp,) : nothing     # sparse twin of the linear block, if any
    x = newvec()
    upload!(x, x0, 0)
    ineq && generate_initial_y!(x, idata)                                                  # :179-182
    obj_values = Float64[]
    xnew, g, d, newton_d = newvec(), newvec(), newvec(), newvec()
    Jct = c! isa DeviceConstraints ? c!.Jct : DeviceMatrix(ctx, n, m)      # device-resident classes own their (mostly constant) Jct
    tmp_m = DeviceVector(ctx, max(m, 1))
    tmp_w = ineq ? DeviceVector(ctx, n) : nothing
    lamy_kkt = ineq ? DeviceVector(ctx, n) : nothing
    hx = ineq ? DeviceVector(ctx, n) : nothing
    cval = zeros(m)
    λ_kkt = zeros(m)
    λ_dev = DeviceVector(ctx, max(m, 1))
    term_cond = f_tol
    prev_grad_norm = 0.0
    diagonal_hessian = has_hess_diag(hess_lag_vec!)
    # The basis Z (src/optimize.jl:191), ProjCGWork (:214) and the operator diagonal the fused iteration reads beside them: allocated TOGETHER,
    # by trial over pairs of candidate allocations (FINDINGS.md 6)
    # ... unless the basis can stay in FACTORED form U = Jct W (FINDINGS.md 5.3): no Z at all, the tangent setup skips its basis-forming product,
    # and the work vectors are placed against Jct
    # (the library says whether this context can run projcg! without Z for this Jct: one-pass kernels on, shape inside their limits, or a
    # sparse twin the nonzero path covers; otherwise Z is materialised and every path has its two-pass form)
    # (a tridiagonal Hessian sent through the callback path -- DeviceOptions.tridiagonal_one_pass off -- needs the materialised basis)
    tri_callback = diagonal_hessian && hess_offdiag(hess_lag_vec!) !== nothing && !ctx.options.tridiagonal_one_pass
    factored_basis = ctx.options.factored_basis && diagonal_hessian && !tri_callback && 4 <= m <= 1024 && factored_basis_supported(ctx, Jct, jsp === nothing ? C_NULL : jsp.h)
    # allocation by trial pays after several hundred projected-CG iterations; a Lagrangian Hessian that is a multiple of I (config 3) ends every
    # truncated-Newton solve after one: such a run takes its first allocations
    saved_tries = ctx.options.placement_tries
    few_cg = scalar_hessian(hess_lag_vec!) && !ineq
    few_cg && saved_tries > 1 && set_placement!(ctx, 1)
    if factored_basis
        vs = vectors_placed(ctx, Jct, ineq ? 0 : n, 5; N=ineq ? n : 0)
        projcgwork = ProjCGWork(vs[1], vs[2], vs[4], DeviceVector(ctx, 3 * max(m, 1) + 8), vs[5])
        a_placed = vs[3]
        idecomp = InequalityDecomp(ctx, n, m, Jct, nothing)
    elseif m > 0
        Zp, vs = basis_and_vectors_placed(ctx, n, m, 5; N=ineq ? n : 0)
        projcgwork = ProjCGWork(vs[1], vs[2], vs[4], DeviceVector(ctx, 3 * max(m, 1) + 8), vs[5])
        a_placed = vs[3]
        idecomp = InequalityDecomp(ctx, n, m, Jct, Zp)
    else
        projcgwork = ProjCGWork(x, m)
        a_placed = nothing
        idecomp = InequalityDecomp(ctx, n, m, Jct)
    end
    ctx.options.placement_tries != saved_tries && set_placement!(ctx, saved_tries)
    # the tangent step with fewer passes (lfpsqp_tangent_step): plain factored basis over dense gradients, truncated-Newton steps on
    # (with bounds: the stacked form of the same pass; not over a matrix view, not for a class whose Hessian term needs A*λ)
    nonlinear_class = c! isa DeviceConstraints && c!.ew !== nothing          # (its Jct may be a view of A, its Hessian term may need A*λ)
    fuse_tangent = factored_basis && param.do_newton && jsp === nothing && m > 0 && ctx.options.fused_tangent_step && !(ineq && nonlinear_class)
    tri_off = diagonal_hessian ? hess_offdiag(hess_lag_vec!) : nothing
    if tri_off !== nothing
        ineq && error("a tridiagonal Hessian with bounds: pass hess_lag_vec! as a function (the generic path)")
        ctx.options.tridiagonal_one_pass || (fuse_tangent = false)     # (the callback path starts its solves itself)
        (haskey(VIEW_KEEP, Jct) || ctx.nranks > 1) && (fuse_tangent = false)  # (lfpsqp_projcg_tridiag refuses a matrix view / several ranks: callback path, its own start)
    end                                                       # (the tangent step still hands projcg! r0 and U'r0; never its folded initial projection)
    ineq_rhs = (fuse_tangent && ineq) ? DeviceVector(ctx, n) : nothing
    Jtd = zeros(max(m, 1)); Utd = zeros(max(m, 1))
    Ggram = fuse_tangent ? zeros(m, m) : nothing              # the factorisation's Gram matrix: U'U = W'GW for the tangent step
    idecomp.W = m > 0 ? zeros(m, m) : nothing                 # ksvd!'s small factor: Z == Jct*W
    jsp === nothing || (idecomp.Jsp = jsp.h)                  # sparse twin: the stacked basis is applied on the nonzeros too
    Z, Σ, Vt = idecomp.Z, idecomp.Σ, idecomp.Vt
    ineqproject = ineq ? InequalityDecompProject(idecomp) : nothing
    a_diag = diagonal_hessian ? (a_placed === nothing ? newvec() : a_placed) : nothing
    # general Hessian with bounds: augmented_hess_lag_vec! (src/inequality_helper.jl:144-158) on the stacked vectors
    aug_src = (!diagonal_hessian && ineq) ? DeviceVector(ctx, n) : nothing
    aug_x = (!diagonal_hessian && ineq) ? DeviceVector(ctx, n) : nothing
    aug_h = (!diagonal_hessian && ineq) ? DeviceVector(ctx, n) : nothing
    aug_d = (!diagonal_hessian && ineq) ? StackedVector(ctx, n) : nothing
    zero_hx = (!diagonal_hessian && ineq) ? DeviceVector(ctx, n) : nothing
    function newton_apply!(dest::DeviceVector, src::DeviceVector)
        if !ineq
            hess_lag_vec!(dest, src, x, λ_dev)
        else
            check(ctx, c_augmented_diag(ctx.h, zero_hx.h, lamy_kkt.h, Ref(cineq(idata)), aug_d.h))   # [2 λy q ; 2 λy s]
            copy_range!(aug_src, 0, src, 0, n); copy_range!(aug_x, 0, x, 0, n)
            hess_lag_vec!(aug_h, aug_src, aug_x, λ_dev)                                                  # H src_x
            vmul!(dest, aug_d, src)
            check(ctx, c_vec_fill_range(ctx.h, aug_d.h, Int64(aug_d.hs), Int64(n), 0.0))             # reuse aug_d as [H src_x ; 0]
            copy_range!(aug_d, 0, aug_h, 0, n)
            axpby!(1.0, aug_d, 1.0, dest)
        end
        return dest
    end
    nr = NR(nothing, Σ, Vt, param.ϵ_c, param.maxiter_retract, ineq, idata)
    pp = ProjPenalty(c! isa DeviceConstraints ? c! : jac!, m, m, param.μ0, param.ϵ_c, param.maxiter_retract, param.maxiter_pcg,
                     ProjPenaltyWork(x, m, n, ineq; against=((m > 0 && !few_cg) ? Jct : nothing)), ineq, idecomp, idata)
    set_nr_batch_mode!(ctx, ctx.options.ls_batch_matrix_cores)
    armijo_work = ArmijoWork(x)
    exact_work = (param.linesearch == exact && !param.disable_linesearch) ? ExactLinesearchWork(x) : nothing
    i = 0
    f_diff = Inf; step_diff = Inf; kkt_diff = Inf
    fval = f(x)
    push!(obj_values, fval)
    if m > 0
        c! isa DeviceConstraints ? c!(cval, x) : c!(cval, download(x, n, 0))
    end
    disp = param.disp == iter && ctx.rank == 0
    if disp
        print_iter_header()
        print_first_line(fval, m > 0 ? maximum(abs, cval) : 0.0)
    end
    noise = nothing
    prev_rank = -1                                                                      # rank of the previous iteration's factorisation
    cval_current = m > 0                                                                # cval == c(x): after the evaluation above and after every accepted retraction
    while true
        grad!(g, x)                                                                     # :259
        waxpby!(d, -1.0, g, 0.0, g)                                                     # :262
        if param.β > 0                                                                  # :264-273 (noise in the descent direction)
            noise === nothing && (noise = newvec())
            z = randn(ineq ? 2n : n)
            ineq ? upload2!(noise, z) : upload!(noise, z)
            axpby!(param.t_β > 0 ? param.β * max(1 - i / param.t_β, 0.0) : param.β, noise, 1.0, d)
        end
        ineq && inequality_gradient!(idecomp, x, idata)                                 # :277
        rank = m
        init_fold = false
        if m > 0
            # :283-284 (the device keeps only Jct); a device-resident class skips the re-evaluation of c(x) when cval holds it already
            (cval_current && jac! isa DeviceConstraints) ? jac!(jac!, Jct, nothing, x) : jac!(Jct, cval, x)
            vprev = (i > 0 && prev_rank == m && ctx.options.warm_factorize) ? copy(Vt) : nothing
            if fuse_tangent && ineq                                                     # ... with bounds: Jct'(sx .* dx + sy .* dy), the m-part of Q'd
                check(ctx, c_ineq_rhs(ctx.h, d.h, idecomp.Dx.h, idecomp.Dy.h, ineq_rhs.h))
                rank = ksvd!(Jct, Z, Σ, Vt; w2=idecomp.sx, ϵ_rank=param.ϵ_rank, W=idecomp.W, Vt_prev=vprev, rhs=ineq_rhs, Jte=Jtd, G=Ggram)
            elseif fuse_tangent                                                         # Jct'd rides with the Gram pass (d is final before jac! runs)
                rank = ksvd!(Jct, Z, Σ, Vt; ϵ_rank=param.ϵ_rank, W=idecomp.W, Vt_prev=vprev, rhs=d, Jte=Jtd, G=Ggram)                 # :286-302
            else
                rank = ksvd!(Jct, Z, Σ, Vt; w2=ineq ? idecomp.sx : nothing, ϵ_rank=param.ϵ_rank, W=idecomp.W, Jsp=jsp, Vt_prev=vprev) # :286-302
            end
            prev_rank = rank
            if fuse_tangent && rank >= 1
                # :305-343, :366-381 and src/projcg.jl:56-59 in one pass: d projected, λ_kkt, the Hessian diagonal completed, r0 = -d and U'r0 left
                # in projcgwork for projcg!(...; start_given=true)
                idecomp.rank = rank
                Ub = ineq ? ineqproject : DeviceBasis(nothing, rank, (Jct, idecomp.W))
                hdst = ineq ? hx : a_diag                                               # where the objective's part goes (bounds: the x-half alone)
                dss = Ref{Float64}(0.0)
                cons_part = nothing
                if has_hess_split(hess_lag_vec!)
                    hess_diag_objective!(hess_lag_vec!, hdst, x)
                    cons_part = hess_constraints(hess_lag_vec!)
                else                                                                    # a diagonal Hessian without the split needs λ_kkt first
                    th = zeros(m)
                    th[1:rank] .= (idecomp.W[:, 1:rank]' * Jtd[1:m]) ./ Σ[1:rank]
                    λ_kkt .= Vt' * th
                    hess_diag!(hess_lag_vec!, hdst, x, λ_kkt)
                end
                # the fold of projcg!'s initial projection: only where the Gram matrix resolves I - U'U (full rank, cond^2 <= 10)
                init_fold = tri_off === nothing && rank == m && Σ[1]^2 <= 10.0 * Σ[m]^2
                GC.@preserve Ub cons_part begin
                    cref = cons_part === nothing ? nothing : Ref(ccons(cons_part))
                    iref = ineq ? Ref(cineq(idata)) : nothing
                    # (LFPSQP_TANGENT_INIT_PROJCG: the pass is projcg!'s initial projection as well, src/projcg.jl:58-62; projcg! starts projected)
                    check(ctx, c_tangent_step(ctx.h, Ref(cbasis(Ub)), Σ, Vt, Int64(m), Jtd, Ggram, d.h,
                                              cref === nothing ? Ptr{CConstraints}(C_NULL) : Base.unsafe_convert(Ptr{CConstraints}, cref), x.h, a_diag.h,
                                              iref === nothing ? Ptr{CIneqData}(C_NULL) : Base.unsafe_convert(Ptr{CIneqData}, iref),
                                              ineq ? hx.h : C_NULL, ineq ? idecomp.S.h : C_NULL, ineq ? lamy_kkt.h : C_NULL,
                                              Ref(cwork(projcgwork)), init_fold ? LFPSQP_TANGENT_INIT_PROJCG : Cint(0), Utd, λ_kkt, dss))
                end
            elseif !ineq                                                                # :305-308
                Ub = jsp === nothing ? (Z === nothing ? DeviceBasis(nothing, rank, (Jct, idecomp.W)) : DeviceBasis(Z, rank)) : DeviceBasis(Z, rank, (Jct, idecomp.W), jsp.h)
                mul!(tmp_m, adjoint(Ub), d)
                mul!(d, Ub, tmp_m, -1.0, 1.0)
            end
        end
        idecomp.rank = rank
        if ineq && !(fuse_tangent && m > 0 && rank >= 1)                                # :312-318
            q_gemv_t!(tmp_w, tmp_m, ineqproject, d)
            q_gemv_n!(d, ineqproject, tmp_w, tmp_m, -1.0, 1.0)
        end
        kkt_diff = norm(d, Inf)                                                         # :320
        pp.rank = rank
        steptype = 0; tn_iter = 0; tn_res = 0.0
        if m > 0 && !(fuse_tangent && rank >= 1)                                        # :331-343 (the tangent step returned λ_kkt otherwise)
            th = download(tmp_m, m)
            th[1:rank] ./= Σ[1:rank]
            th[rank+1:m] .= 0.0
            λ_kkt .= Vt' * th
            upload!(λ_dev, λ_kkt)
        end
        if ineq && !(fuse_tangent && m > 0 && rank >= 1)                                # calculate_λ_kkt!, inequality_helper.jl:286-308
            check(ctx, c_calculate_lambda_y(ctx.h, Jct.h, Int64(m), λ_dev.h, idecomp.Dx.h, idecomp.S.h, tmp_w.h, lamy_kkt.h))
        end
        if f_diff <= param.ϵ_f                                                          # :347-359
            term_cond = f_tol; break
        elseif step_diff <= param.ϵ_x
            term_cond = x_tol; break
        elseif i >= param.maxiter
            term_cond = max_iter; break
        elseif kkt_diff <= param.ϵ_kkt
            term_cond = kkt_tol; break
        end
        if param.do_newton                                                              # :364-390
            Qview = ineq ? ineqproject : (jsp === nothing ? (Z === nothing ? DeviceBasis(nothing, rank, (Jct, idecomp.W)) : DeviceBasis(Z, rank)) : DeviceBasis(Z, rank, (Jct, idecomp.W), jsp.h))   # sparse twin: projcg! on the nonzeros
            grad_norm = norm(d)
            tol = param.tn_κ * min(1, grad_norm / prev_grad_norm) * grad_norm           # :375-378 (prev = 0: ratio Inf => factor 1)
            prev_grad_norm = grad_norm
            nglob = ineq ? 2 * n_global : n_global
            if diagonal_hessian
                fused_now = fuse_tangent && rank >= 1
                if fused_now                                                            # (a_diag was completed by the tangent step)
                elseif ineq
                    hess_diag!(hess_lag_vec!, hx, x, λ_kkt)
                    check(ctx, c_augmented_diag(ctx.h, hx.h, lamy_kkt.h, Ref(cineq(idata)), a_diag.h))
                else
                    hess_diag!(hess_lag_vec!, a_diag, x, λ_kkt)
                end
                if tri_off !== nothing && !ctx.options.tridiagonal_one_pass
                    Atri = TridiagonalOperator(0.0, a_diag, tri_off)
                    tn_iter, tn_res = projcg!(newton_d, nothing, (dest, src) -> mul!(dest, Atri, src), Qview, d, nothing; tol=tol, maxit=param.tn_maxiter,
                                              work=projcgwork, n_global=nglob)
                elseif tri_off !== nothing
                    tn_iter, tn_res = projcg!(newton_d, nothing, TridiagonalOperator(0.0, a_diag, tri_off), Qview, d, nothing; tol=tol, maxit=param.tn_maxiter,
                                              work=projcgwork, n_global=nglob, start_given=fused_now)
                else
                    tn_iter, tn_res = projcg!(newton_d, nothing, DiagOperator(0.0, a_diag), Qview, d, nothing; tol=tol, maxit=param.tn_maxiter,
                                              work=projcgwork, n_global=nglob, start_projected=fused_now && init_fold, start_given=fused_now && !init_fold)
                end
            else
                tn_iter, tn_res = projcg!(newton_d, nothing, newton_apply!, Qview, d, nothing; tol=tol, maxit=param.tn_maxiter,
                                          work=projcgwork, n_global=nglob)
            end
            if dot(newton_d, d) > 0.0                                                   # :386
                copyto!(d, newton_d)
                steptype = 1
            end
        end
        retract_method = ineq ? YRetract(idata) : Euclidean()                           # :396-412
        mtype = 0
        if m > 0
            if rank == m && !param.do_project_retract
                nr.U = ineq ? ineqproject : DeviceBasis(Z, rank, (Jct, idecomp.W))
                retract_method, mtype = nr, 0
            else
                retract_method, mtype = pp, 1
            end
        end
        if param.linesearch == armijo || param.disable_linesearch
            flag, iter1, iter2, newf, f_diff, step_diff, α = armijo!(xnew, x, n, d, g, f, fval, retract_method, cval, c!, param, armijo_work)
        else
            flag, iter1, iter2, newf, f_diff, step_diff, α = exact_linesearch!(xnew, x, n, d, f, fval, retract_method, cval, c!, param, exact_work)
        end
        copyto!(x, xnew)                                                                # :424-427
        # (Armijo's accepted trial is the last one retracted: cval == c!(xnew); the exact search returns a saved best point, cval is the last trial's)
        cval_current = flag == 0 && m > 0 && (param.linesearch == armijo || param.disable_linesearch)
        fval = newf
        push!(obj_values, fval)
        disp && print_iter(i + 1, fval, m > 0 ? maximum(abs, cval) : 0.0, f_diff, step_diff, steptype, tn_iter, tn_res, mtype, iter1, iter2, α, flag)
        i += 1
        if param.callback !== nothing && i % param.callback_period == 0
            param.callback(i, x)
            cval_current = false            # the callback holds the live iterate: if it edits x, cval is stale (the reference's jac! recomputes it, :283)
        end
    end
    (i == param.maxiter && disp) && println("Warning: Maximum # of outer iterations reached")
    return download(x, n, 0), obj_values, λ_kkt, TerminationInfo(term_cond, f_diff, step_diff, kkt_diff, i)
end

# ---- device-resident problem class of BASELINE configs 2-5 --------------------------------------------------------------------
# f = ||x - xc||², dense linear equalities J x = b, optional ball x'x <= R2 (an equality with a slack variable exactly as
# src/optimize.jl:23-51 does it) and optional box bounds; f, grad!, c!, jac! and the (diagonal) Lagrangian Hessian run on the device.
struct QuadLinearBallBox
    ctx::HipContext
    n::Int
    m::Int
    p::Int
    ploc::Int                   # 1 on the rank that owns the slack variable
    Jct::DeviceMatrix           # (n + ploc) x (m + p); the slack row and the ball column are managed here
    cons::DeviceConstraints
    R2::Float64
    xl::Union{Nothing,Vector{Float64}}
    xu::Union{Nothing,Vector{Float64}}
    xc::Float64
    n_global::Int
end
function QuadLinearBallBox(ctx::HipContext, n::Int, m::Int, Jct::DeviceMatrix, b::Vector{Float64}; R2=nothing, xl=nothing, xu=nothing, xc::Real=0.0,
                           n_global::Int=n, owns_slack::Bool=true)
    p = R2 === nothing ? 0 : 1
    ploc = owns_slack ? p : 0
    (Jct.n == n + ploc && Jct.m == m + p) || error("Jct must be (n + ploc) x (m + p)")
    cons = DeviceConstraints(Jct, m, m > 0 ? copy(b) : zeros(1), p == 1, R2 === nothing ? 0.0 : Float64(R2), n, ploc == 1 ? n : -1)
    return QuadLinearBallBox(ctx, n, m, p, ploc, Jct, cons, R2 === nothing ? 0.0 : Float64(R2), xl, xu, Float64(xc), n_global)
end
function objective(P::QuadLinearBallBox, x::DeviceVector)
    out = Ref{Float64}(0.0)
    check(P.ctx, c_sumsq_shift(P.ctx.h, x.h, Int64(P.n), P.xc, out))
    return out[]
end
gradient!(P::QuadLinearBallBox, g::DeviceVector, x::DeviceVector) = (check(P.ctx, c_affine_head(P.ctx.h, 2.0, x.h, -2.0 * P.xc, Int64(P.n), g.h)); g)
# diag of ∇²f + Σ λ_i ∇²c_i: 2 (+ 2 λ_ball) on the user's variables, 0 on the slack
function hess_diag!(P::QuadLinearBallBox, hx::DeviceVector, x::DeviceVector, λ::Vector{Float64})
    fill_range!(hx, 0, P.n, 2.0 + (P.p == 1 ? 2.0 * λ[P.m+1] : 0.0))
    P.ploc == 1 && fill_range!(hx, P.n, 1, 0.0)
    return hx
end
function hess_diag_objective!(P::QuadLinearBallBox, hx::DeviceVector, x::DeviceVector)
    fill_range!(hx, 0, P.n, 2.0)
    P.ploc == 1 && fill_range!(hx, P.n, 1, 0.0)
    return hx
end
hess_constraints(P::QuadLinearBallBox) = P.cons
scalar_hessian(P::QuadLinearBallBox) = P.p == 0
function optimize(P::QuadLinearBallBox, x0::Vector{Float64}, param::LFPSQPParams=LFPSQPParams())
    x0a, xl, xu = x0, P.xl, P.xu
    if P.p == 1                                                                   # the slack transformation of src/optimize.jl:23-36
        tmp = upload!(DeviceVector(P.ctx, P.n), x0)
        out = Ref{Float64}(0.0)
        check(P.ctx, c_sumsq_shift(P.ctx.h, tmp.h, Int64(P.n), 0.0, out))          # global x0'x0 (all-reduced over the shards)
        xl = P.xl === nothing ? fill(-Inf, P.n) : P.xl
        xu = P.xu === nothing ? fill(Inf, P.n) : P.xu
        if P.ploc == 1
            x0a = vcat(x0, out[] - P.R2); xl = vcat(xl, -Inf); xu = vcat(xu, 0.0)
        end
    end
    x, obj, λ, info = optimize_core(P.ctx, x -> objective(P, x), (g, x) -> gradient!(P, g, x), P.cons, (J, cv, x) -> jac!(P.cons, J, cv, x), P,
                                    x0a, xl, xu, P.m + P.p, param; n_global=P.n_global + P.p)
    return x[1:P.n], obj, λ, info
end

# ---- a second device-resident class: separable objective f(x) = Σ φ(x_i - c_i; a_i) (lfpsqp_separable) under the same constraints ----
# kind 0: a t², 1: a t⁴ + t², 2: a (√(1 + t²) - 1); f, grad! and the diagonal Lagrangian Hessian φ''(x_i) + 2 λ_ball are elementwise kernels
struct SeparableLinearBallBox
    base::QuadLinearBallBox
    kind::Int
    a::DeviceVector
    c::DeviceVector
    lamvec::DeviceVector        # scratch: 2 λ_ball on the user's variables
end
function SeparableLinearBallBox(ctx::HipContext, n::Int, m::Int, Jct::DeviceMatrix, b::Vector{Float64}, kind::Int, a::Vector{Float64}, c::Vector{Float64}; kw...)
    base = QuadLinearBallBox(ctx, n, m, Jct, b; kw...)
    return SeparableLinearBallBox(base, kind, upload!(DeviceVector(ctx, n), a), upload!(DeviceVector(ctx, n), c), DeviceVector(ctx, n + base.ploc))
end
function objective(P::SeparableLinearBallBox, x::DeviceVector)
    out = Ref{Float64}(0.0)
    check(P.base.ctx, c_separable(P.base.ctx.h, Cint(P.kind), Cint(0), P.a.h, 0.0, P.c.h, 0.0, x.h, Int64(P.base.n), C_NULL, out))
    return out[]
end
function gradient!(P::SeparableLinearBallBox, g::DeviceVector, x::DeviceVector)
    check(P.base.ctx, c_separable(P.base.ctx.h, Cint(P.kind), Cint(1), P.a.h, 0.0, P.c.h, 0.0, x.h, Int64(P.base.n), g.h, Ptr{Float64}(C_NULL)))
    P.base.ploc == 1 && fill_range!(g, P.base.n, 1, 0.0)
    return g
end
function hess_diag!(P::SeparableLinearBallBox, hx::DeviceVector, x::DeviceVector, λ::Vector{Float64})
    B = P.base
    check(B.ctx, c_separable(B.ctx.h, Cint(P.kind), Cint(2), P.a.h, 0.0, P.c.h, 0.0, x.h, Int64(B.n), hx.h, Ptr{Float64}(C_NULL)))
    if B.p == 1
        fill_range!(P.lamvec, 0, B.n, 2.0 * λ[B.m+1])
        check(B.ctx, c_axpby(B.ctx.h, 1.0, P.lamvec.h, 1.0, hx.h))
    end
    B.ploc == 1 && fill_range!(hx, B.n, 1, 0.0)
    return hx
end
function hess_diag_objective!(P::SeparableLinearBallBox, hx::DeviceVector, x::DeviceVector)
    B = P.base
    check(B.ctx, c_separable(B.ctx.h, Cint(P.kind), Cint(2), P.a.h, 0.0, P.c.h, 0.0, x.h, Int64(B.n), hx.h, Ptr{Float64}(C_NULL)))
    B.ploc == 1 && fill_range!(hx, B.n, 1, 0.0)
    return hx
end
hess_constraints(P::SeparableLinearBallBox) = P.base.cons
function optimize(P::SeparableLinearBallBox, x0::Vector{Float64}, param::LFPSQPParams=LFPSQPParams())
    B = P.base
    x0a, xl, xu = x0, B.xl, B.xu
    if B.p == 1
        tmp = upload!(DeviceVector(B.ctx, B.n), x0)
        out = Ref{Float64}(0.0)
        check(B.ctx, c_sumsq_shift(B.ctx.h, tmp.h, Int64(B.n), 0.0, out))
        xl = B.xl === nothing ? fill(-Inf, B.n) : B.xl
        xu = B.xu === nothing ? fill(Inf, B.n) : B.xu
        if B.ploc == 1
            x0a = vcat(x0, out[] - B.R2); xl = vcat(xl, -Inf); xu = vcat(xu, 0.0)
        end
    end
    x, obj, λ, info = optimize_core(B.ctx, x -> objective(P, x), (g, x) -> gradient!(P, g, x), B.cons, (J, cv, x) -> jac!(B.cons, J, cv, x), P,
                                    x0a, xl, xu, B.m + B.p, param; n_global=B.n_global + B.p)
    return x[1:B.n], obj, λ, info
end

# ---- separable objective plus a CHAIN term: f(x) = Σ φ(x_i - c_i; a_i) + κ/2 Σ_{i<n} (x_{i+1} - x_i)² under dense linear equalities ----------------
# (the kind of objective whose Hessian the reference reaches only through hess_lag_vec!, src/autodiff_generators.jl:72-107).  The Lagrangian
# Hessian is TRIDIAGONAL: diagonal φ''(x_i) + κ deg_i (deg = 1 at the two ends, 2 inside), couplings -κ.  optimize_core finds the couplings
# through hess_offdiag and runs its truncated-Newton solves on the one-pass solver (lfpsqp_projcg_tridiag).  One rank, no ball, no bounds.
struct ChainSeparableLinear
    sep::SeparableLinearBallBox
    κ::Float64
    deg::DeviceVector           # κ .* degree of the path graph
    off::DeviceVector           # -κ (entry n is ignored)
    tmp::DeviceVector
end
function ChainSeparableLinear(ctx::HipContext, n::Int, m::Int, Jct::DeviceMatrix, b::Vector{Float64}, kind::Int, a::Vector{Float64}, c::Vector{Float64}; κ::Float64=1.0)
    ctx.nranks == 1 || error("chain objective: one rank (the couplings would cross the shard boundaries)")
    sep = SeparableLinearBallBox(ctx, n, m, Jct, b, kind, a, c)
    deg = fill(2.0 * κ, n); deg[1] = deg[n] = n > 1 ? κ : 0.0
    return ChainSeparableLinear(sep, κ, upload!(DeviceVector(ctx, n), deg), upload!(DeviceVector(ctx, n), fill(-κ, n)), DeviceVector(ctx, n))
end
laplacian(P::ChainSeparableLinear) = TridiagonalOperator(0.0, P.deg, P.off)         # κ L, L = the path graph's Laplacian
function objective(P::ChainSeparableLinear, x::DeviceVector)
    mul!(P.tmp, laplacian(P), x)
    return objective(P.sep, x) + 0.5 * dot(x, P.tmp)
end
function gradient!(P::ChainSeparableLinear, g::DeviceVector, x::DeviceVector)
    gradient!(P.sep, g, x)
    mul!(P.tmp, laplacian(P), x)
    check(x.ctx, c_axpby(x.ctx.h, 1.0, P.tmp.h, 1.0, g.h))
    return g
end
function hess_diag!(P::ChainSeparableLinear, hx::DeviceVector, x::DeviceVector, λ::Vector{Float64})
    hess_diag!(P.sep, hx, x, λ)
    check(x.ctx, c_axpby(x.ctx.h, 1.0, P.deg.h, 1.0, hx.h))
    return hx
end
function hess_diag_objective!(P::ChainSeparableLinear, hx::DeviceVector, x::DeviceVector)
    hess_diag_objective!(P.sep, hx, x)
    check(x.ctx, c_axpby(x.ctx.h, 1.0, P.deg.h, 1.0, hx.h))
    return hx
end
hess_constraints(P::ChainSeparableLinear) = P.sep.base.cons
hess_offdiag(P::ChainSeparableLinear) = P.off
function optimize(P::ChainSeparableLinear, x0::Vector{Float64}, param::LFPSQPParams=LFPSQPParams())
    B = P.sep.base
    x, obj, λ, info = optimize_core(B.ctx, x -> objective(P, x), (g, x) -> gradient!(P, g, x), B.cons, (J, cv, x) -> jac!(B.cons, J, cv, x), P,
                                    x0, nothing, nothing, B.m, param; n_global=B.n_global)
    return x[1:B.n], obj, λ, info
end

# ---- device-resident class with NONLINEAR equalities: separable objective under ElementwiseConstraints, optional box bounds ----------
struct SeparableElementwiseBox
    ctx::HipContext
    cons::DeviceConstraints     # from ElementwiseConstraints
    kind::Int
    a::DeviceVector
    c::DeviceVector
    xl::Union{Nothing,Vector{Float64}}
    xu::Union{Nothing,Vector{Float64}}
    n_global::Int
end
SeparableElementwiseBox(ctx::HipContext, cons::DeviceConstraints, kind::Int, a::Vector{Float64}, c::Vector{Float64}; xl=nothing, xu=nothing,
                        n_global::Int=cons.Jct.n) =
    SeparableElementwiseBox(ctx, cons, kind, upload!(DeviceVector(ctx, cons.Jct.n), a), upload!(DeviceVector(ctx, cons.Jct.n), c), xl, xu, n_global)
function objective(P::SeparableElementwiseBox, x::DeviceVector)
    out = Ref{Float64}(0.0)
    check(P.ctx, c_separable(P.ctx.h, Cint(P.kind), Cint(0), P.a.h, 0.0, P.c.h, 0.0, x.h, Int64(P.cons.Jct.n), C_NULL, out))
    return out[]
end
gradient!(P::SeparableElementwiseBox, g::DeviceVector, x::DeviceVector) =
    (check(P.ctx, c_separable(P.ctx.h, Cint(P.kind), Cint(1), P.a.h, 0.0, P.c.h, 0.0, x.h, Int64(P.cons.Jct.n), g.h, Ptr{Float64}(C_NULL))); g)
function hess_diag!(P::SeparableElementwiseBox, hx::DeviceVector, x::DeviceVector, λ::Vector{Float64})
    check(P.ctx, c_separable(P.ctx.h, Cint(P.kind), Cint(2), P.a.h, 0.0, P.c.h, 0.0, x.h, Int64(P.cons.Jct.n), hx.h, Ptr{Float64}(C_NULL)))
    return hess_diag!(P.cons, hx, x, λ)
end
hess_diag_objective!(P::SeparableElementwiseBox, hx::DeviceVector, x::DeviceVector) =
    (check(P.ctx, c_separable(P.ctx.h, Cint(P.kind), Cint(2), P.a.h, 0.0, P.c.h, 0.0, x.h, Int64(P.cons.Jct.n), hx.h, Ptr{Float64}(C_NULL))); hx)
hess_constraints(P::SeparableElementwiseBox) = P.cons
function optimize(P::SeparableElementwiseBox, x0::Vector{Float64}, param::LFPSQPParams=LFPSQPParams())
    return optimize_core(P.ctx, x -> objective(P, x), (g, x) -> gradient!(P, g, x), P.cons, (J, cv, x) -> jac!(P.cons, J, cv, x), P,
                         x0, P.xl, P.xu, P.cons.m_lin, param; n_global=P.n_global)
end

# ---- the reference's method table with arbitrary HOST callables (src/optimize.jl:13, 83, 88, 107, 112, 119) --------------------
# Iterates are downloaded for every user call: plumbing / small problems (config 1), not the 1e7-variable configs.
# The AD generators (src/autodiff_generators.jl) stay where they are: pass their outputs (grad!, jac!, hess_lag_vec!) here.
function optimize(ctx::HipContext, f, grad!, c!, jac!, hess_lag_vec!, x0::Vector{Float64}, xl, xu, m::Int, param::LFPSQPParams=LFPSQPParams())
    n = length(x0)
    f_dev(x) = Float64(f(download(x, n, 0)))
    function grad_dev!(g, x)
        gh = zeros(n)
        grad!(gh, download(x, n, 0))
        upload!(g, gh, 0)
    end
    function jac_dev!(Jct, cval, x)
        J = zeros(m, n)
        jac!(J, cval, download(x, n, 0))
        upload!(Jct, Matrix(J'))
    end
    function hlv_dev!(dest, src, x, λ_dev)
        out = zeros(n)
        hess_lag_vec!(out, download(src, n, 0), download(x, n, 0), download(λ_dev, max(m, 1))[1:m])
        upload!(dest, out, 0)
    end
    return optimize_core(ctx, f_dev, grad_dev!, c!, m > 0 ? jac_dev! : nothing, hlv_dev!, x0, xl, xu, m, param)
end
# unconstrained / equalities only / equalities + bounds: the derivatives come from the maintainer's AD generators
optimize(ctx::HipContext, f, grad!, hess_lag_vec!, x0::Vector{Float64}, param::LFPSQPParams=LFPSQPParams()) =
    optimize(ctx, f, grad!, nothing, nothing, hess_lag_vec!, x0, nothing, nothing, 0, param)                                   # src/optimize.jl:112
optimize(ctx::HipContext, f, grad!, c!, jac!, hess_lag_vec!, x0::Vector{Float64}, m::Int, param::LFPSQPParams=LFPSQPParams()) =
    optimize(ctx, f, grad!, c!, jac!, hess_lag_vec!, x0, nothing, nothing, m, param)                                             # :107
# general inequalities dl <= d(x) <= du through slack variables (:13-71): n -> n + p, m -> m + p, result truncated (:68)
function optimize(ctx::HipContext, f, grad!, c!, jac_c!, d!, jac_d!, hess_lag_vec!, dl::Vector{Float64}, du::Vector{Float64}, x0::Vector{Float64},
                  xl, xu, m::Int, p::Int, param::LFPSQPParams=LFPSQPParams())
    (d! === nothing || p == 0) && return optimize(ctx, f, grad!, c!, jac_c!, hess_lag_vec!, x0, xl, xu, m, param)                  # :88
    length(dl) == length(du) == p || error("Bound vectors dl and du must be of size p")
    n = length(x0)
    xl = xl === nothing ? fill(-Inf, n) : xl
    xu = xu === nothing ? fill(Inf, n) : xu
    x0_aux = vcat(x0, zeros(p))
    d!(view(x0_aux, n+1:n+p), x0)
    f_aux(x) = f(x[1:n])
    function c_aux!(cval, x)
        m > 0 && c!(view(cval, 1:m), x[1:n])
        d!(view(cval, m+1:m+p), x[1:n])
        cval[m+1:m+p] .-= x[n+1:n+p]
        return cval
    end
    grad_aux!(g, x) = (grad!(view(g, 1:n), x[1:n]); g[n+1:end] .= 0.0; g)
    function jac_aux!(J, cval, x)
        J .= 0.0
        m > 0 && jac_c!(view(J, 1:m, 1:n), view(cval, 1:m), x[1:n])
        jac_d!(view(J, m+1:m+p, 1:n), view(cval, m+1:m+p), x[1:n])
        cval[m+1:m+p] .-= x[n+1:n+p]
        for k in 1:p
            J[m+k, n+k] = -1.0
        end
    end
    hlv_aux!(dest, src, x, λ) = (hess_lag_vec!(view(dest, 1:n), src[1:n], x[1:n], λ); dest[n+1:end] .= 0.0; dest)
    x, obj, λ, info = optimize(ctx, f_aux, grad_aux!, c_aux!, jac_aux!, hlv_aux!, x0_aux, vcat(xl, dl), vcat(xu, du), m + p, param)
    return x[1:n], obj, λ, info
end
# d(x) <= 0 (:83)
optimize(ctx::HipContext, f, grad!, c!, jac_c!, d!, jac_d!, hess_lag_vec!, x0::Vector{Float64}, xl, xu, m::Int, p::Int, param::LFPSQPParams=LFPSQPParams()) =
    optimize(ctx, f, grad!, c!, jac_c!, d!, jac_d!, hess_lag_vec!, fill(-Inf, p), zeros(p), x0, xl, xu, m, p, param)

export HipContext, HipError, DeviceOptions, DeviceVector, StackedVector, DeviceMatrix, SparseMatrix, spmv_t!, spmv_n!, to_dense!, DeviceBasis, DiagOperator, LowRankOperator, InequalityData, InequalityDecomp,
       InequalityDecompProject, ProjCGWork, DeviceConstraints, NR, ProjPenalty, ProjPenaltyWork, Euclidean, YRetract, ArmijoWork,
       ExactLinesearchWork, LFPSQPParams, TerminationInfo, QuadLinearBallBox, SeparableLinearBallBox, ChainSeparableLinear, TridiagonalOperator, SeparableElementwiseBox, ElementwiseConstraints,
       sin_system_constraints, sphere_system_constraints, clone, rowscale!, set_placement!, basis_and_vectors_placed, vectors_placed, placement_info, upload!, download, upload2!, download2, projcg!, retract!,
       retract_nr_batch!, pcg!, ProjPrecondition, ksvd!, armijo!, exact_linesearch!, optimize, optimize_core, hess_diag!, jac!, comm_unique_id, comm_init!, comm_p2p_export, comm_init_p2p!,
       shard_range, sync

end # module
