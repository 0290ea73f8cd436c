# LFPSQPHip.jl -- the binding a maintainer of ksil/LFPSQP.jl adds to run the inner-loop hot path
# (projcg!, retract!, the tangent setup, the bound operators) on an MI355X through liblfpsqp_hip.so.
#
# NOT EXECUTED in this repository's CI: no `julia` binary exists in the build image (SURVEY.md §0),
# so this file is syntax-reviewed only; every ccall below is exercised through the identical C ABI
# by the Python ctypes mirror (lfpsqp.jl_amd/_capi.py) in tests/.  See INTEGRATION.md.
#
# Usage inside LFPSQP.jl:   include("LFPSQPHip.jl"); using .LFPSQPHip
#   ctx = HipContext(0)
#   U   = DeviceBasis(upload(ctx, Umatrix))            # or factorize!(ctx, Jct_dev, Z_dev)
#   A   = DiagOperator(2.0)                            # hess_lag_vec! = 2v  (a0*I + diag(dg))
#   i, nr = projcg!(x_dev, λ_dev, A, U, b_dev, nothing; tol=tol, maxit=maxit, work=work)
# i.e. the call sites of src/optimize.jl:381 / src/linesearch.jl:52 stay as they are; only the
# array / operator types change and Julia's dispatch selects the methods defined here.
module LFPSQPHip

using LinearAlgebra
import LinearAlgebra: mul!, dot, norm

const lib = get(ENV, "LFPSQP_HIP_LIB", joinpath(@__DIR__, "..", "lfpsqp.jl_amd", "lib", "liblfpsqp_hip.so"))

struct HipError <: Exception
    code::Cint
    msg::String
end

mutable struct HipContext
    h::Ptr{Cvoid}
    function HipContext(device::Integer=0)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:lfpsqp_ctx_create, lib), Cint, (Cint, Ref{Ptr{Cvoid}}), device, r)
        rc == 0 || throw(HipError(rc, "lfpsqp_ctx_create failed: no usable MI355X (there is no CPU fallback)"))
        ctx = new(r[])
        finalizer(c -> ccall((:lfpsqp_ctx_destroy, lib), Cint, (Ptr{Cvoid},), c.h), ctx)
        return ctx
    end
end

check(ctx::HipContext, rc::Cint) = rc == 0 ? nothing :
    throw(HipError(rc, unsafe_string(ccall((:lfpsqp_last_error, lib), Cstring, (Ptr{Cvoid},), ctx.h))))

# ---- buffers --------------------------------------------------------------------------------
mutable struct DeviceVector <: AbstractVector{Float64}
    ctx::HipContext
    h::Ptr{Cvoid}
    n::Int
end
function DeviceVector(ctx::HipContext, n::Integer)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ctx, ccall((:lfpsqp_vec_alloc, lib), Cint, (Ptr{Cvoid}, Int64, Ref{Ptr{Cvoid}}), ctx.h, n, r))
    v = DeviceVector(ctx, r[], n)
    finalizer(x -> ccall((:lfpsqp_vec_free, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), x.ctx.h, x.h), v)
    return v
end
Base.size(v::DeviceVector) = (v.n,)
Base.length(v::DeviceVector) = v.n
upload!(v::DeviceVector, host::Vector{Float64}) =
    (check(v.ctx, ccall((:lfpsqp_vec_upload, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Float64}, Int64), v.ctx.h, v.h, 0, host, length(host))); v)
function download(v::DeviceVector)
    host = Vector{Float64}(undef, v.n)
    check(v.ctx, ccall((:lfpsqp_vec_download, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Float64}, Int64), v.ctx.h, v.h, 0, host, v.n))
    return host
end

mutable struct DeviceMatrix <: AbstractMatrix{Float64}
    ctx::HipContext
    h::Ptr{Cvoid}
    n::Int
    m::Int
end
function DeviceMatrix(ctx::HipContext, n::Integer, m::Integer)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ctx, ccall((:lfpsqp_mat_alloc, lib), Cint, (Ptr{Cvoid}, Int64, Int64, Ref{Ptr{Cvoid}}), ctx.h, n, m, r))
    M = DeviceMatrix(ctx, r[], n, m)
    finalizer(x -> ccall((:lfpsqp_mat_free, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), x.ctx.h, x.h), M)
    return M
end
Base.size(M::DeviceMatrix) = (M.n, M.m)
upload!(M::DeviceMatrix, host::Matrix{Float64}) =      # Julia matrices are column-major: zero-copy layout match
    (check(M.ctx, ccall((:lfpsqp_mat_upload, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Int64),
                        M.ctx.h, M.h, 0, size(host, 2), host, size(host, 1))); M)

# ---- BLAS-1/2 on device arrays: the generic-dispatch surface projcg!/pcg! are written against --
# replaces src/la_helper.jl:36-44 (kgemv!) and the mul!/dot/norm calls of src/projcg.jl, src/retractions.jl
struct DeviceBasis            # view(U, :, 1:rank), src/optimize.jl:370
    Z::DeviceMatrix
    ncols::Int
end
DeviceBasis(Z::DeviceMatrix) = DeviceBasis(Z, Z.m)
struct DeviceBasisAdjoint
    U::DeviceBasis
end
Base.adjoint(U::DeviceBasis) = DeviceBasisAdjoint(U)
Base.adjoint(Ut::DeviceBasisAdjoint) = Ut.U

mul!(y::DeviceVector, U::DeviceBasis, t::DeviceVector, a::Number=1.0, b::Number=0.0) =
    (check(y.ctx, ccall((:lfpsqp_gemv_n, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Float64, Ptr{Cvoid}, Float64, Ptr{Cvoid}),
                        y.ctx.h, U.Z.h, U.ncols, a, t.h, b, y.h)); y)
mul!(t::DeviceVector, Ut::DeviceBasisAdjoint, v::DeviceVector) =
    (check(t.ctx, ccall((:lfpsqp_gemv_t, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
                        t.ctx.h, Ut.U.Z.h, Ut.U.ncols, v.h, t.h)); t)
function dot(x::DeviceVector, y::DeviceVector)
    r = Ref{Float64}(0.0)
    check(x.ctx, ccall((:lfpsqp_dot, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}), x.ctx.h, x.h, y.h, r))
    return r[]
end
function norm(x::DeviceVector, p::Real=2)
    r = Ref{Float64}(0.0)
    if p == Inf
        check(x.ctx, ccall((:lfpsqp_amax, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}), x.ctx.h, x.h, r))
    else
        check(x.ctx, ccall((:lfpsqp_nrm2, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}), x.ctx.h, x.h, r))
    end
    return r[]
end
axpby!(a::Number, x::DeviceVector, b::Number, y::DeviceVector) =
    (check(y.ctx, ccall((:lfpsqp_axpby, lib), Cint, (Ptr{Cvoid}, Float64, Ptr{Cvoid}, Float64, Ptr{Cvoid}), y.ctx.h, a, x.h, b, y.h)); y)

# ---- tangent setup: replaces ksvd! (src/la_helper.jl:8-34), call sites src/optimize.jl:291/293 -----
# W (optional m×m): the small factor with Z == Jct*W; a DeviceBasis that carries (Jct, W) lets the Newton retraction
# stream Jct once per step instead of Z and Jct.
function ksvd!(Jct::DeviceMatrix, Z::DeviceMatrix, Σ::Vector{Float64}, Vt::Matrix{Float64}; w2=nothing, ϵ_rank::Float64=1e-10,
               W::Union{Nothing,Matrix{Float64}}=nothing)
    rank = Ref{Int64}(0)
    check(Jct.ctx, ccall((:lfpsqp_factorize, lib), Cint,
                         (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Int64}, Float64),
                         Jct.ctx.h, Jct.h, w2 === nothing ? C_NULL : w2.h, Z.h, Σ, Vt, W === nothing ? C_NULL : W, rank, ϵ_rank))
    return Int(rank[])
end

# ---- projcg! (src/projcg.jl:40-121): fused device solve selected by dispatch on (A, U) ---------------
struct DiagOperator            # A = a0*I + diag(dg): the LinearMap of src/optimize.jl:228-230 for diagonal Hessians
    a0::Float64
    dg::Union{Nothing,DeviceVector}
end
DiagOperator(a0::Real) = DiagOperator(Float64(a0), nothing)

struct CDiagOp;  a0::Float64; dg::Ptr{Cvoid}; end
struct CBasis;   Z::Ptr{Cvoid}; ncols::Int64; Dx::Ptr{Cvoid}; Dy::Ptr{Cvoid}; sx::Ptr{Cvoid}; sy::Ptr{Cvoid}; A::Ptr{Cvoid}; W::Ptr{Float64}; end
CBasis(Z, ncols, Dx, Dy, sx, sy) = CBasis(Z, ncols, Dx, Dy, sx, sy, C_NULL, C_NULL)      # generator unknown
struct CWork;    g::Ptr{Cvoid}; d::Ptr{Cvoid}; rp::Ptr{Cvoid}; Utr::Ptr{Cvoid}; w::Ptr{Cvoid}; end

struct ProjCGWork              # ProjCGWork(n, m), src/projcg.jl:1-11 (three n-vectors suffice on the device)
    g::DeviceVector; d::DeviceVector; rp::DeviceVector; Utr::DeviceVector
end
ProjCGWork(ctx::HipContext, n::Int, m::Int) =
    ProjCGWork(DeviceVector(ctx, n), DeviceVector(ctx, n), DeviceVector(ctx, n), DeviceVector(ctx, max(m, 1)))

function projcg!(x::DeviceVector, λ::Union{Nothing,DeviceVector}, A::DiagOperator, U::DeviceBasis, b::DeviceVector, c::Union{Nothing,DeviceVector};
                 tol::Float64=1e-6, maxit::Int=length(b) + U.ncols, work::ProjCGWork=ProjCGWork(x.ctx, length(b), U.ncols),
                 n_global::Int=length(b))
    iters = Ref{Int64}(0); nr = Ref{Float64}(0.0)
    a = Ref(CDiagOp(A.a0, A.dg === nothing ? C_NULL : A.dg.h))
    u = Ref(CBasis(U.Z.h, U.ncols, C_NULL, C_NULL, C_NULL, C_NULL))
    w = Ref(CWork(work.g.h, work.d.h, work.rp.h, work.Utr.h, C_NULL))
    check(x.ctx, ccall((:lfpsqp_projcg, lib), Cint,
                       (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{CDiagOp}, Ref{CBasis}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Int64, Int64, Cint,
                        Ref{CWork}, Ref{Int64}, Ref{Float64}),
                       x.ctx.h, x.h, λ === nothing ? C_NULL : λ.h, a, u, b.h, c === nothing ? C_NULL : c.h, tol, maxit, n_global,
                       λ === nothing ? 0 : 1, w, iters, nr))
    return Int(iters[]), nr[]          # (i, nr) exactly like the reference; nr == Inf and λ .== NaN on negative curvature
end

# ---- Newton retraction (src/retractions.jl:75-177) with device-resident linear + ball constraints ------
struct CConstraints
    Jct::Ptr{Cvoid}; m_lin::Int64; b::Ptr{Float64}; has_ball::Cint; R2::Float64; n_x::Int64; slack_row::Int64
end
struct DeviceNR                # NR(U, Σ, Vt, tol, maxiter, work, ineq, idata) of the reference
    U::DeviceBasis; Σ::Vector{Float64}; Vt::Matrix{Float64}; tol::Float64; maxiter::Int
    Jct::DeviceMatrix; m_lin::Int; b::Vector{Float64}; has_ball::Bool; R2::Float64; n_x::Int; slack_row::Int
    W::Union{Nothing,Matrix{Float64}}      # ksvd!'s W (U.Z == Jct*W): one matrix stream per Newton step; nothing = two streams
end
function retract!(cval::Vector{Float64}, xnew::DeviceVector, c!, xtilde::DeviceVector, x::DeviceVector, method::DeviceNR)
    flag = Ref{Cint}(0); iters = Ref{Int64}(0)
    GC.@preserve method begin
        u = Ref(method.W === nothing ? CBasis(method.U.Z.h, method.U.ncols, C_NULL, C_NULL, C_NULL, C_NULL) :
                CBasis(method.U.Z.h, method.U.ncols, C_NULL, C_NULL, C_NULL, C_NULL, method.Jct.h, pointer(method.W)))
        cons = Ref(CConstraints(method.Jct.h, method.m_lin, pointer(method.b), method.has_ball ? 1 : 0, method.R2, method.n_x, method.slack_row))
        check(x.ctx, ccall((:lfpsqp_retract_nr, lib), Cint,
                           (Ptr{Cvoid}, Ref{CBasis}, Ptr{Float64}, Ptr{Float64}, Int64, Ref{CConstraints}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
                            Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Int64, Ptr{Float64}, Ref{Cint}, Ref{Int64}),
                           x.ctx.h, u, method.Σ, method.Vt, length(method.Σ), cons, C_NULL, C_NULL, C_NULL,
                           xtilde.h, x.h, xnew.h, method.tol, method.maxiter, cval, flag, iters))
    end
    return Int(flag[]), Int(iters[]), 0      # (flag, iter1, iter2) as in the reference
end

# ---- ProjPenalty, the reference's DEFAULT retraction (src/retractions.jl:265-441) and its pcg! (:179-246) ------
struct CPPWork
    r::Ptr{Cvoid}; p::Ptr{Cvoid}; z::Ptr{Cvoid}; dx::Ptr{Cvoid}; g::Ptr{Cvoid}; tmp_m::Ptr{Cvoid}
    tmp_w::Ptr{Cvoid}; h::Ptr{Cvoid}; DxS::Ptr{Cvoid}; DyS::Ptr{Cvoid}; ones::Ptr{Cvoid}; zeros::Ptr{Cvoid}
end
struct DevicePPWork            # ProjPenaltyWork(m, n, m_ineq, n_ineq) without bounds
    r::DeviceVector; p::DeviceVector; z::DeviceVector; dx::DeviceVector; g::DeviceVector; tmp_m::DeviceVector
end
DevicePPWork(ctx::HipContext, n::Int, m::Int) =
    DevicePPWork((DeviceVector(ctx, n) for _ in 1:5)..., DeviceVector(ctx, max(m, 1)))
struct DevicePP                # ProjPenalty(jac!, U, Σ, Vt, rank, μ0, tol, maxiter, maxiter_pcg, work, ineq, idecomp, idata)
    Jct::DeviceMatrix; m_lin::Int; b::Vector{Float64}; has_ball::Bool; R2::Float64; n_x::Int; slack_row::Int
    μ0::Float64; tol::Float64; maxiter::Int; maxiter_pcg::Int; work::DevicePPWork
end
function retract!(cval::Vector{Float64}, xnew::DeviceVector, c!, xtilde::DeviceVector, x::DeviceVector, pp::DevicePP)
    flag = Ref{Cint}(0); iters = Ref{Int64}(0); pcg_iters = Ref{Int64}(0)
    w = pp.work
    wc = Ref(CPPWork(w.r.h, w.p.h, w.z.h, w.dx.h, w.g.h, w.tmp_m.h, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL))
    GC.@preserve pp begin
        cons = Ref(CConstraints(pp.Jct.h, pp.m_lin, pointer(pp.b), pp.has_ball ? 1 : 0, pp.R2, pp.n_x, pp.slack_row))
        check(x.ctx, ccall((:lfpsqp_retract_pp, lib), Cint,
                           (Ptr{Cvoid}, Ref{CConstraints}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid},
                            Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Int64, Int64,
                            Ref{CPPWork}, Ptr{Float64}, Ref{Cint}, Ref{Int64}, Ref{Int64}),
                           x.ctx.h, cons, C_NULL, C_NULL, C_NULL, pp.Jct.h, length(cval), C_NULL, C_NULL, C_NULL, C_NULL,
                           xtilde.h, x.h, xnew.h, pp.μ0, pp.tol, pp.maxiter, pp.maxiter_pcg, wc, cval, flag, iters, pcg_iters))
    end
    return Int(flag[]), Int(iters[]), Int(pcg_iters[])      # (flag, iter1, iter2) as in the reference
end
# pcg!(μ, J, no_precondition, x, r, p, z, tmp_m, tol, maxiter) with J' = Jct given as a DeviceBasis
function pcg!(μ::Float64, Jt::DeviceBasis, x::DeviceVector, r::DeviceVector, p::DeviceVector, z::DeviceVector,
              tmp_m::DeviceVector, tol::Float64, maxiter::Int)
    flag = Ref{Cint}(0); iters = Ref{Int64}(0)
    u = Ref(CBasis(Jt.Z.h, Jt.ncols, C_NULL, C_NULL, C_NULL, C_NULL))
    check(x.ctx, ccall((:lfpsqp_pcg, lib), Cint,
                       (Ptr{Cvoid}, Float64, Ref{CBasis}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
                        Float64, Int64, Ref{Cint}, Ref{Int64}),
                       x.ctx.h, μ, u, x.h, r.h, p.h, z.h, C_NULL, tmp_m.h, tol, maxiter, flag, iters))
    return Int(flag[]), Int(iters[])
end

# ---- multi-GPU: one Julia process per GPU (e.g. under MPI.jl); rank 0 creates the id and broadcasts it --
function comm_unique_id(ctx::HipContext)
    id = Vector{UInt8}(undef, 128)
    check(ctx, ccall((:lfpsqp_comm_unique_id, lib), Cint, (Ptr{Cvoid}, Ptr{UInt8}), ctx.h, id))
    return id
end
comm_init!(ctx::HipContext, rank::Integer, nranks::Integer, id::Vector{UInt8}) =
    check(ctx, ccall((:lfpsqp_comm_init_rccl, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{UInt8}), ctx.h, rank, nranks, id))

export HipContext, DeviceVector, DeviceMatrix, DeviceBasis, DiagOperator, ProjCGWork, DeviceNR, DevicePP, DevicePPWork,
       upload!, download, projcg!, retract!, pcg!, ksvd!, comm_unique_id, comm_init!

end # module

# ---- optional: several trial points of one Armijo search retracted together --------------------------------------------
# armijo! (src/linesearch.jl:32-89) tries x + α d, x + α s d, ... one after another; when retractions fail (100 Newton
# iterations each) the trial points are independent and can share every pass over Jct.  A maintainer who wants this
# adds a lookahead to armijo! (see lfpsqp.jl_amd/linesearch.py::armijo_ for the bookkeeping that keeps the accepted step
# and all counts those of the one-by-one search) on top of this call.
function retract_batch!(cvals::Matrix{Float64}, xnews::Vector{DeviceVector}, xtildes::Vector{DeviceVector}, x::DeviceVector, method::DeviceNR)
    nb = length(xnews)
    flags = zeros(Cint, nb); iters = zeros(Int64, nb)
    xt = [v.h for v in xtildes]; xn = [v.h for v in xnews]
    GC.@preserve method xt xn begin
        u = Ref(CBasis(method.U.Z.h, method.U.ncols, C_NULL, C_NULL, C_NULL, C_NULL, method.Jct.h, pointer(method.W)))
        cons = Ref(CConstraints(method.Jct.h, method.m_lin, pointer(method.b), method.has_ball ? 1 : 0, method.R2, method.n_x, method.slack_row))
        check(x.ctx, ccall((:lfpsqp_retract_nr_batch, lib), Cint,
                           (Ptr{Cvoid}, Ref{CBasis}, Ptr{Float64}, Ptr{Float64}, Int64, Ref{CConstraints}, Ptr{Cvoid}, Cint, Ptr{Ptr{Cvoid}},
                            Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Float64, Int64, Ptr{Float64}, Ptr{Cint}, Ptr{Int64}),
                           x.ctx.h, u, method.Σ, method.Vt, length(method.Σ), cons, C_NULL, nb, xt, x.h, xn, method.tol, method.maxiter,
                           cvals, flags, iters))       # cvals is m × nb (column b = cval of trial b)
    end
    return [(Int(flags[b]), Int(iters[b]), 0) for b in 1:nb]
end
