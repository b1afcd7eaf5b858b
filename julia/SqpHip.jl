# SqpHip.jl -- binding of libsqphip.so (MI355X hot path) into SqpSolver.jl.
#
# `include` this file inside module SqpSolver after src/algorithms/subproblem_JuMP.jl
# (src/algorithms/subproblem.jl:30-31).  It adds `QpHip <: AbstractSubOptimizer`, the sub-problem seat of
# src/algorithms/subproblem.jl:1 as QpJuMP fills it (subproblem_JuMP.jl:127-183 QP, :185-244 LP phase, :283-347
# L1QP, :352-393 feasibility restoration, :398-429 infeasibility problem), plus thin wrappers of the merit path and of
# the multi-GPU status gather.  The only edits to existing code are three one-line branches in
# src/algorithms/sqp_trust_region.jl (:264, :314, :341; see "dispatch" at the end of this file).
#
# No Julia toolchain exists in the image this project is built in: this file is UNEXERCISED.  The call sequence and
# the struct layouts are exercised through the same C ABI by tests/c_abi_smoke.c (plain C, dlopen) and by the Python
# binding sqpsolver.jl_amd/host.py, which mirrors this file one to one.

const LIBSQPHIP = get(ENV, "SQPHIP_LIB", "libsqphip")

# mirrors sqphip_options (include/sqphip.h); field order and types are the ABI
struct SqpHipOptions
    tol_direction::Cdouble; tol_residual::Cdouble; tol_infeas::Cdouble
    init_mu::Cdouble; max_mu::Cdouble; tr_size::Cdouble
    rho::Cdouble; eta::Cdouble; tau::Cdouble; min_alpha::Cdouble
    max_iter::Int32; use_soc::Int32; literal_quirks::Int32
    ipm_tol::Cdouble
    ipm_max_iter::Int32; ipm_phase1::Int32; device::Int32; ipm_corrector::Int32
    kkt_condense::Int32; kkt_tile_order::Int32; kkt_mode::Int32; ipm_warm_start::Int32
end

function sqphip_default_options()
    o = Ref{SqpHipOptions}()
    ccall((:sqphip_default_options, LIBSQPHIP), Cvoid, (Ref{SqpHipOptions},), o)
    return o[]
end

# options of the context from SqpSolver's Parameters (src/parameters.jl:1-30); everything else keeps its default
function SqpHipOptions(par; device::Integer = 0, literal_quirks::Integer = 1)
    d = sqphip_default_options()
    return SqpHipOptions(par.tol_direction, par.tol_residual, par.tol_infeas, par.init_mu, par.max_mu, par.tr_size,
                         par.rho, par.eta, par.tau, par.min_alpha, par.max_iter, par.use_soc ? 1 : 0, literal_quirks,
                         d.ipm_tol, d.ipm_max_iter, d.ipm_phase1, device, d.ipm_corrector, d.kkt_condense,
                         d.kkt_tile_order, d.kkt_mode, d.ipm_warm_start)
end

mutable struct QpHip{T,Tv<:AbstractArray{T},Tm<:AbstractMatrix{T}} <: AbstractSubOptimizer
    ctx::Ptr{Cvoid}
    data::QpData{T,Tv,Tm}
    sqp                          # back-reference: the COO values dE / h_val live in the SQP struct
end

_check(ctx, rc) = rc == 0 || error("libsqphip error $rc: " *
    unsafe_string(ccall((:sqphip_last_error, LIBSQPHIP), Cstring, (Ptr{Cvoid},), ctx)))

function QpHip(sqp::AbstractSqpOptimizer; device::Integer = 0)
    pr = sqp.problem
    opt = Ref(SqpHipOptions(sqp.options; device = device))
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:sqphip_create, LIBSQPHIP), Cint,
               (Ref{Ptr{Cvoid}}, Int64, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Int64, Ptr{Int64},
                Ptr{Int64}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{SqpHipOptions}, Int32),
               ctx, pr.n, pr.m, pr.num_linear_constraints,
               length(sqp.j_row), sqp.j_row, sqp.j_col,          # 1-based COO as stored at sqp_trust_region.jl:41-54
               length(sqp.h_row), sqp.h_row, sqp.h_col,
               pr.x_L, pr.x_U, pr.g_L, pr.g_U, opt, 1)
    rc == 0 || error("sqphip_create failed ($rc)")
    qp = QpHip(ctx[], QpData(sqp), sqp)
    finalizer(q -> ccall((:sqphip_destroy, LIBSQPHIP), Cvoid, (Ptr{Cvoid},), q.ctx), qp)
    return qp
end

create_model!(qp::QpHip, Δ) = nothing        # subproblem_JuMP.jl:36-125 has no counterpart: the context is the model

# modes of include/sqphip.h: 0 QP, 1 FR, 2 SOC, 3 LP phase, 4 L1QP, 5 INFEAS
function _solve(qp::QpHip, mode::Integer, x_k, Δ, μ = 1.0)
    n, m = length(qp.data.c), length(qp.data.c_lb)
    p, λ, mU, mL = zeros(n), zeros(m), zeros(n), zeros(n)
    slack = zeros(2m); st = Ref{Int32}(0)
    hval = isnothing(qp.data.Q) ? C_NULL : pointer(qp.sqp.h_val)
    rc = ccall((:sqphip_qp_solve, LIBSQPHIP), Cint,
               (Ptr{Cvoid}, Int32, Ptr{Cdouble}, Cdouble, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
                Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Int32}),
               qp.ctx, mode, x_k, Δ, μ, qp.data.c, qp.data.b, qp.sqp.dE, hval, p, λ, mU, mL, slack, st)
    _check(qp.ctx, rc)
    p_slack = Dict(i => [slack[i], slack[m+i]] for i in (qp.data.num_linear_constraints+1):m)
    return p, λ, mU, mL, p_slack, MOI.TerminationStatusCode(st[])      # the 6-tuple of QpJuMP (collect_solution!, :514-563)
end

sub_optimize!(qp::QpHip, x_k, Δ)         = _solve(qp, 0, x_k, Δ)                 # subproblem_JuMP.jl:127-183
sub_optimize_FR!(qp::QpHip, x_k, Δ)      = _solve(qp, 1, x_k, Δ)                 # :352-393
sub_optimize_L1QP!(qp::QpHip, x_k, Δ, μ) = _solve(qp, 4, x_k, Δ, μ)              # :283-347
function sub_optimize_infeas(qp::QpHip, x_k, Δ)                                   # :398-429
    p, _, _, _, sl, st = _solve(qp, 5, x_k, Δ)
    return p, st == MOI.LOCALLY_SOLVED ? sum(sum, values(sl)) : Inf
end
function sub_optimize_lp(qp::QpHip, x_k)                                          # :185-244, result is the absolute point
    x, λ, mU, mL, _, st = _solve(qp, 3, x_k, Inf)
    return x, λ, mU, mL, st
end

# ---- merit path: one ccall each, host vectors in, scalar out --------------------------------------------------------
function hip_norm_violations(qp::QpHip, E, x, p = 1)                               # common.jl:54-77
    out = Ref{Cdouble}(0); code = p == Inf ? 0 : Int(p)
    _check(qp.ctx, ccall((:sqphip_norm_violations, LIBSQPHIP), Cint,
                         (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Int32, Ref{Cdouble}), qp.ctx, E, x, code, out))
    return out[]
end
function hip_KT_residuals(qp::QpHip, df, λ, mU, mL, dE)                             # common.jl:14-23
    out = Ref{Cdouble}(0)
    _check(qp.ctx, ccall((:sqphip_kt_residuals, LIBSQPHIP), Cint,
                         (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Cdouble}),
                         qp.ctx, df, λ, mU, mL, dE, out))
    return out[]
end
function hip_compute_phi(qp::QpHip, f_trial, E_trial, x_trial, μ, fr::Bool)        # sqp.jl:170-183
    out = Ref{Cdouble}(0)
    _check(qp.ctx, ccall((:sqphip_compute_phi, LIBSQPHIP), Cint,
                         (Ptr{Cvoid}, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Int32, Ref{Cdouble}),
                         qp.ctx, f_trial, E_trial, x_trial, μ, fr ? 1 : 0, out))
    return out[]
end
function hip_compute_qmodel(qp::QpHip, x, p, df, E, dE, h_val, μ, with_step::Bool)  # sqp_trust_region.jl:487-508
    out = Ref{Cdouble}(0)
    _check(qp.ctx, ccall((:sqphip_compute_qmodel, LIBSQPHIP), Cint,
                         (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
                          Cdouble, Int32, Ref{Cdouble}), qp.ctx, x, p, df, E, dE, h_val, μ, with_step ? 1 : 0, out))
    return out[]
end
# compute_derivative(sqp): sqp.jl:190-213 over merit.jl:13-17 (μ scalar or vector, restoration branch)
function hip_compute_derivative(qp::QpHip, df, p, E, μ, fr::Bool, slack)
    out = Ref{Cdouble}(0)
    μs, μv = μ isa Number ? (Cdouble(μ), C_NULL) : (0.0, pointer(μ))
    _check(qp.ctx, ccall((:sqphip_compute_derivative_full, LIBSQPHIP), Cint,
                         (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}, Int32, Ptr{Cdouble},
                          Ref{Cdouble}), qp.ctx, df, p, E, μs, μv, fr ? 1 : 0, slack, out))
    return out[]
end
function hip_tr_update(ared, pred, Δ, pnorm, Δmax, tol_direction)                   # sqp_trust_region.jl:529-538, :574-577
    acc = Ref{Int32}(0); Δn = Ref{Cdouble}(0)
    ccall((:sqphip_tr_update, LIBSQPHIP), Cint, (Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Ref{Int32}, Ref{Cdouble}),
          ared, pred, Δ, pnorm, Δmax, tol_direction, acc, Δn)
    return acc[] == 1, Δn[]
end

# ---- multi-GPU: status gather over RCCL inside the library (include/sqphip.h, sqphip_gather_status) -----------------
# rank 0: id = hip_comm_unique_id(); ship the 128 bytes to the other ranks (Distributed / MPI); every rank:
# hip_comm_init(qp, id, world, rank); then hip_gather_status(qp, total) after each batch of outer iterations.
function hip_comm_unique_id()
    id = zeros(UInt8, 128)
    rc = ccall((:sqphip_comm_unique_id, LIBSQPHIP), Cint, (Ptr{UInt8},), id)
    rc == 0 || error("sqphip_comm_unique_id failed ($rc)")
    return id
end
hip_comm_available() = ccall((:sqphip_comm_available, LIBSQPHIP), Cint, ()) == 1     # ask on every rank before hip_comm_init
hip_comm_init(qp::QpHip, id::Vector{UInt8}, world::Integer, rank::Integer) =
    _check(qp.ctx, ccall((:sqphip_comm_init, LIBSQPHIP), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Int32, Int32), qp.ctx, id, world, rank))
function hip_gather_status(qp::QpHip, total::Integer)
    ret, it, done = zeros(Int32, total), zeros(Int32, total), zeros(Int32, total)
    _check(qp.ctx, ccall((:sqphip_gather_status, LIBSQPHIP), Cint, (Ptr{Cvoid}, Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}),
                         qp.ctx, total, ret, it, done))
    return ret, it, done
end

# ---- batched ACOPF runs on the device (include/sqphip.h; no reference counterpart): a context created with the structure
# of the ACP or ACR layout, `hip_acopf_attach` (rectangular = true: ACRPowerModel, examples/acopf/opf.jl:46), per-instance
# data, `hip_sqp_run`; or a scenario queue when there are more scenarios than slots ------------------------------------
function hip_acopf_attach(ctx::Ptr{Cvoid}, nb, ng, nl, f_bus::Vector{Int32}, t_bus::Vector{Int32}, gen_bus::Vector{Int32},
                          bal_ptr::Vector{Int32}, bal_colP::Vector{Int32}, bal_colQ::Vector{Int32},
                          bal_coef::Vector{Float64}, ref_bus; rectangular::Bool = false)
    args = (ctx, Int32(nb), Int32(ng), Int32(nl), f_bus, t_bus, gen_bus, bal_ptr, bal_colP, bal_colQ, bal_coef, Int32(ref_bus))
    T = (Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Cdouble}, Int32)
    rc = rectangular ? ccall((:sqphip_acopf_attach_acr, LIBSQPHIP), Cint, T, args...) :
                       ccall((:sqphip_acopf_attach, LIBSQPHIP), Cint, T, args...)
    _check(ctx, rc)
end
# the W-space model of examples/acopf/acwr.jl (ACWRPowerModel): bus pairs i < j, pair / orientation of every branch,
# tan of the pairs' angle limits (0-based indices, as everything at this boundary below the COO structure)
hip_acopf_attach_acwr(ctx::Ptr{Cvoid}, nb, ng, nl, f_bus::Vector{Int32}, t_bus::Vector{Int32}, gen_bus::Vector{Int32},
                      bal_ptr::Vector{Int32}, bal_colP::Vector{Int32}, bal_colQ::Vector{Int32}, bal_coef::Vector{Float64},
                      ref_bus, bp_i::Vector{Int32}, bp_j::Vector{Int32}, br_bp::Vector{Int32}, br_sig::Vector{Float64},
                      bp_tmin::Vector{Float64}, bp_tmax::Vector{Float64}) =
    _check(ctx, ccall((:sqphip_acopf_attach_acwr, LIBSQPHIP), Cint,
                      (Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32},
                       Ptr{Cdouble}, Int32, Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
                      ctx, nb, ng, nl, f_bus, t_bus, gen_bus, bal_ptr, bal_colP, bal_colQ, bal_coef, ref_bus,
                      length(bp_i), bp_i, bp_j, br_bp, br_sig, bp_tmin, bp_tmax))
hip_sqp_run(ctx::Ptr{Cvoid}, max_outer::Integer = 0) =
    _check(ctx, ccall((:sqphip_sqp_run, LIBSQPHIP), Cint, (Ptr{Cvoid}, Int32), ctx, max_outer))
hip_stream_begin(ctx::Ptr{Cvoid}, n_scenarios::Integer) =
    _check(ctx, ccall((:sqphip_sqp_stream_begin, LIBSQPHIP), Cint, (Ptr{Cvoid}, Int32), ctx, n_scenarios))
hip_stream_set(ctx::Ptr{Cvoid}, s::Integer, xL, xU, gL, gU, ohm, c2, c1, x0) =        # s is 0-based
    _check(ctx, ccall((:sqphip_sqp_stream_set, LIBSQPHIP), Cint,
                      (Ptr{Cvoid}, Int32, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
                       Ptr{Cdouble}, Ptr{Cdouble}), ctx, s, xL, xU, gL, gU, ohm, c2, c1, x0))
hip_stream_run(ctx::Ptr{Cvoid}) = _check(ctx, ccall((:sqphip_sqp_stream_run, LIBSQPHIP), Cint, (Ptr{Cvoid},), ctx))
function hip_stream_get(ctx::Ptr{Cvoid}, s::Integer, n::Integer)
    x = zeros(n); obj = Ref{Cdouble}(0.0); st = Ref{Cint}(0); it = Ref{Cint}(0)
    _check(ctx, ccall((:sqphip_sqp_stream_get, LIBSQPHIP), Cint, (Ptr{Cvoid}, Int32, Ptr{Cdouble}, Ref{Cdouble}, Ref{Cint}, Ref{Cint}),
                      ctx, s, x, obj, st, it))
    return x, obj[], st[], it[]
end
# work of a batched run by instance (sub-problems, interior-point iterations, factorisations)
function hip_sqp_work(ctx::Ptr{Cvoid}, batch::Integer)
    qp, ipm, fac = zeros(Int64, batch), zeros(Int64, batch), zeros(Int64, batch)
    _check(ctx, ccall((:sqphip_sqp_work, LIBSQPHIP), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}), ctx, qp, ipm, fac))
    return qp, ipm, fac
end

# ---- dispatch ----------------------------------------------------------------------------------------------------------
# The user selects the device path with
#     optimizer_with_attributes(SqpSolver.Optimizer, "external_optimizer" => SqpSolver.SqpHipBackend, ...)
# (`external_optimizer::Union{Nothing,DataType,...}`, src/parameters.jl:6-7: a struct type is a DataType).  Three call
# sites of src/algorithms/sqp_trust_region.jl then branch on `uses_hip(sqp)`, one line each (INTEGRATION.md has the diff):
#     :264  sub_optimize_lp!(sqp)   -> uses_hip(sqp) && return sub_optimize_lp_hip!(sqp)
#     :314  sub_optimize!(sqp)      -> uses_hip(sqp) && return sub_optimize_hip!(sqp)
#     :341  sub_optimize_soc!(sqp)  -> uses_hip(sqp) && return sub_optimize_soc_hip!(sqp)
struct SqpHipBackend end

uses_hip(sqp::AbstractSqpOptimizer) = sqp.options.external_optimizer === SqpHipBackend

# the context is created on first use and lives as long as the SqpTR object (sqp.optimizer, as upstream at :315-320)
function _hip_seat!(sqp::AbstractSqpTrOptimizer)
    if isnothing(sqp.optimizer)
        sqp.optimizer = QpHip(sqp)
    end
    return sqp.optimizer
end

# sub_optimize!(sqp), sqp_trust_region.jl:314-331
function sub_optimize_hip!(sqp::AbstractSqpTrOptimizer)
    qp = _hip_seat!(sqp)
    qp.data = QpData(sqp)                                   # as :322; c, b are read from data, dE / h_val from sqp
    if sqp.feasibility_restoration
        return sub_optimize_FR!(qp, sqp.x, sqp.Δ)
    else
        return sub_optimize!(qp, sqp.x, sqp.Δ)
    end
end

# sub_optimize_soc!(sqp), sqp_trust_region.jl:341-360: the same seat in mode 2 with b = E_soc.  Upstream builds a FRESH
# QpData around E_soc (:344-355); QpData(sqp) aliases b === sqp.E (sqp.jl:66-79), so writing E_soc through it would
# overwrite the constraint values of the current iterate.
function sub_optimize_soc_hip!(sqp::AbstractSqpTrOptimizer)
    qp = _hip_seat!(sqp)
    sqp.problem.eval_g(sqp.x + sqp.p, sqp.E_soc)
    sqp.E_soc -= sqp.Jacobian * sqp.p
    qp.data = QpData(MOI.MIN_SENSE, sqp.Hessian, sqp.df, sqp.Jacobian, sqp.E_soc, sqp.problem.g_L, sqp.problem.g_U,
                     sqp.problem.x_L, sqp.problem.x_U, sqp.problem.num_linear_constraints)
    p, _, _, _, _, _ = _solve(qp, 2, sqp.x, sqp.Δ)
    sqp.p_soc .= sqp.p .+ p
    return nothing
end

# sub_optimize_lp!(sqp), sqp_trust_region.jl:264-304: the projection of the start point onto the linear rows and the
# variable bounds (subproblem_JuMP.jl:185-244) through the seat's mode 3; the Jacobian values travel in sqp.dE
# (eval_Jacobian!, sqp.jl:111-117), the result is the absolute point, small entries are dropped as upstream (:299-302)
function sub_optimize_lp_hip!(sqp::AbstractSqpTrOptimizer)
    sqp.f = sqp.problem.eval_f(sqp.x)
    sqp.problem.eval_grad_f(sqp.x, sqp.df)
    eval_Jacobian!(sqp)
    qp = _hip_seat!(sqp)
    qp.data = QpData(sqp)
    sqp.x, sqp.lambda, sqp.mult_x_U, sqp.mult_x_L, sqp.sub_status = sub_optimize_lp(qp, sqp.x)
    dropzeros!(sqp.x)
    dropzeros!(sqp.lambda)
    dropzeros!(sqp.mult_x_U)
    dropzeros!(sqp.mult_x_L)
    return
end
