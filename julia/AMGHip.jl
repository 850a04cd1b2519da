# AMGHip.jl — the reference-side binding a maintainer of AlgebraicMultigrid.jl would add to run the
# solve phase on MI355X through libamghip's C ABI (include/amghip.h).
#
# NOT EXECUTED in this repository: there is no Julia toolchain on the build or GPU machines
# (SURVEY.md §0).  It is written against Julia >= 1.6 semantics and the reference's v2.0.0
# sources; the same call sequence is exercised from Python (algebraicmultigrid.jl_amd/device.py),
# which is the tested boundary.  See INTEGRATION.md.
#
# Design: the hierarchy is built by the reference's own setup (`ruge_stuben`,
# `smoothed_aggregation`) unchanged.  `hip(ml)` uploads it once and returns a `MultiLevel` whose
# workspace type parameter TW is `HipWorkspace`; two methods specialised on that type forward
# `_solve!` and the smoother / mul! hooks to the C ABI, so `_solve(ml, b)`, `aspreconditioner(ml)`,
# `ldiv!`, `\`, `solve(A, b, RugeStubenAMG())` keep their exact signatures (multilevel.jl:152-198,
# preconditioner.jl:10-24).
module AMGHip

using AlgebraicMultigrid
using SparseArrays, LinearAlgebra
import AlgebraicMultigrid: MultiLevel, Level, _solve!, Cycle, V, W, F, GaussSeidel, Jacobi, SOR,
                           ForwardSweep, BackwardSweep, SymmetricSweep, Pinv, QRSolver,
                           HermitianSymmetry, NoSymmetry, FastGSSmoother, FastJacobiSmoother,
                           FastSORSmoother

const libamghip = get(ENV, "LIBAMGHIP", "libamghip.so")

struct AMGHipError <: Exception
    rc::Cint
end
Base.showerror(io::IO, e::AMGHipError) =
    print(io, "libamghip: ", unsafe_string(ccall((:amgh_strerror, libamghip), Cstring, (Cint,), e.rc)))
check(rc) = rc == 0 ? nothing : throw(AMGHipError(rc))

# amgh_smoother_t (include/amghip.h)
struct CSmoother
    kind::Int32; sweep::Int32; iter::Int32; pad::Int32; omega::Float64
end
sweepcode(::ForwardSweep) = Int32(0); sweepcode(::BackwardSweep) = Int32(1); sweepcode(::SymmetricSweep) = Int32(2)
csmoother(s::FastGSSmoother{S}) where {S} = CSmoother(1, sweepcode(S()), s.iter, 0, 1.0)
csmoother(s::FastJacobiSmoother) = CSmoother(2, 2, s.iter, 0, Float64(s.ω))
csmoother(s::FastSORSmoother{S}) where {S} = CSmoother(3, sweepcode(S()), s.iter, 0, Float64(s.ω))
# NoSymmetry caches map to the same kernels on the true rows (S == A); see amgh_push_level.

"""Workspace type that marks a MultiLevel as resident on the GPU (replaces MultiLevelWorkspace)."""
mutable struct HipWorkspace
    handle::Ptr{Cvoid}
    function HipWorkspace(h)
        w = new(h)
        finalizer(w -> ccall((:amgh_destroy, libamghip), Cvoid, (Ptr{Cvoid},), w.handle), w)
    end
end
Base.eltype(::HipWorkspace) = Float64

# 0-based int32 CSR arrays of M given Julia's 1-based CSC of M' (CSC arrays of X are CSR arrays of X')
csr_of_transpose(X::SparseMatrixCSC) = (Int32.(X.colptr .- 1), Int32.(X.rowval .- 1), Float64.(X.nzval))
csr(X::SparseMatrixCSC) = csr_of_transpose(copy(X'))
csr(X::Adjoint{<:Any,<:SparseMatrixCSC}) = csr_of_transpose(parent(X))   # lazy adjoint: arrays are already the CSR

"""
    hip(ml::MultiLevel; device = 0) -> MultiLevel

Upload the hierarchy to HBM (amgh_create / amgh_push_level / amgh_set_coarse / amgh_finalize).
"""
function hip(ml::MultiLevel; device::Integer = 0, symmetry = HermitianSymmetry())
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:amgh_create, libamghip), Cint, (Ref{Ptr{Cvoid}}, Cint, Cint), h, device, 1))
    for lev in ml.levels
        A = lev.A
        n, nc = size(lev.P)
        Ar, Ac, Av = csr(A)                                   # true rows: mul!(res, A, x)
        sym = symmetry isa HermitianSymmetry
        # the "fast" smoothers read CSC column i as row i (smoother.jl:81-86): S = CSC arrays as CSR
        S = (sym && !issymmetric(A)) ? csr_of_transpose(A) : nothing
        Pr, Pc, Pv = csr(lev.P); Rr, Rc, Rv = csr(lev.R)
        pre, post = Ref(csmoother(lev.presmoother)), Ref(csmoother(lev.postsmoother))
        GC.@preserve Ar Ac Av S Pr Pc Pv Rr Rc Rv check(ccall((:amgh_push_level, libamghip), Cint,
            (Ptr{Cvoid}, Int64, Int64, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64},
             Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Ref{CSmoother}, Ref{CSmoother}),
            h[], n, nc, Ar, Ac, Av,
            S === nothing ? C_NULL : S[1], S === nothing ? C_NULL : S[2], S === nothing ? C_NULL : S[3],
            Pr, Pc, Pv, Rr, Rc, Rv, pre, post))
    end
    fA = ml.final_A
    n = size(fA, 1)
    fr, fc, fv = csr(fA)
    cs = ml.coarse_solver
    if n <= 2048
        # Pinv: the stored pinv(Matrix(A)) (coarse_solver.jl:11); otherwise the dense inverse standing in
        # for the factorisation solve (coarse_solver.jl:75-81)
        op = cs isa Pinv ? Matrix{Float64}(cs.pinvA) : Matrix{Float64}(inv(Matrix(fA)))
        check(ccall((:amgh_set_coarse, libamghip), Cint,
                    (Ptr{Cvoid}, Int64, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}), h[], n, fr, fc, fv, op))
    else
        # pluggable host coarse solver: the reference's `(cs)(x, b)` protocol through a C callback
        cb = @cfunction($((user, b, x, n) -> begin
                 cs(unsafe_wrap(Array, x, n), unsafe_wrap(Array, b, n)); Cint(0)
             end), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64))
        check(ccall((:amgh_set_coarse_host, libamghip), Cint,
                    (Ptr{Cvoid}, Int64, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Ptr{Cvoid}, Ptr{Cvoid}),
                    h[], n, fr, fc, fv, cb, C_NULL))
    end
    check(ccall((:amgh_finalize, libamghip), Cint, (Ptr{Cvoid},), h[]))
    MultiLevel(ml.levels, ml.final_A, ml.coarse_solver, ml.presmoother, ml.postsmoother, HipWorkspace(h[]))
end

cyclecode(::V) = Cint(0); cyclecode(::W) = Cint(1); cyclecode(::F) = Cint(2)

const HipML = MultiLevel{<:Any,<:Any,<:Any,<:Any,<:Any,<:Any,HipWorkspace}

# _solve!(x, ml, b, cycle; maxiter, abstol, reltol, verbose, log, calculate_residual)  multilevel.jl:158-198
function AlgebraicMultigrid._solve!(x, ml::HipML, b::AbstractVector{Float64}, cycle::Cycle = V();
                                    maxiter::Int = 100, abstol::Real = 0.0, reltol::Real = sqrt(eps(Float64)),
                                    verbose::Bool = false, log::Bool = false, calculate_residual = true, kwargs...)
    hist = zeros(Float64, maxiter + 1)
    iters = Ref{Cint}(0)
    check(ccall((:amgh_solve, libamghip), Cint,
                (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Cint, Cint, Float64, Float64, Cint, Ptr{Float64}, Ref{Cint}),
                ml.workspace.handle, b, x, cyclecode(cycle), maxiter, abstol, reltol, calculate_residual, hist, iters))
    if verbose && calculate_residual
        for i in 1:iters[]
            Printf.@printf "Norm of residual at iteration %6d is %.4e\n" i hist[i]
        end
    end
    log ? (x, hist[1:(calculate_residual ? iters[] + 1 : 1)]) : x
end

# `aspreconditioner(ml)`, `ldiv!`, `\` need no new methods: preconditioner.jl:12-19 calls `_solve!` with
# maxiter = 1, calculate_residual = false, which lands in the method above.  A Krylov loop that wants to
# stay on the device uses amgh_pcg / amgh_precond_apply_d directly:
function cg(ml::HipML, b::Vector{Float64}; cycle::Cycle = V(), maxiter::Int = length(b), abstol = 0.0,
            reltol = sqrt(eps(Float64)))
    x = zeros(length(b)); iters = Ref{Cint}(0)
    check(ccall((:amgh_pcg, libamghip), Cint,
                (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Cint, Cint, Cint, Float64, Float64, Ptr{Float64}, Ref{Cint}),
                ml.workspace.handle, b, x, cyclecode(cycle), 1, maxiter, abstol, reltol, C_NULL, iters))
    x
end

end # module
