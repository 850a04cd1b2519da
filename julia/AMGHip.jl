# AMGHip.jl — the reference-side binding a maintainer of AlgebraicMultigrid.jl would add to run the
# solve phase on MI355X through libamghip's C ABI (include/amghip.h).
#
# A SKETCH: NEVER EXECUTED in this repository — there is no Julia toolchain on the build or GPU machines
# (SURVEY.md §0).  It is written against Julia >= 1.6 semantics and the reference's v2.0.0 sources; the
# same call sequence is exercised from Python (algebraicmultigrid.jl_amd/device.py, sharded.py), which is
# the tested boundary.  See INTEGRATION.md.
#
# Design: the hierarchy is built by the reference's own setup (`ruge_stuben`, `smoothed_aggregation`)
# unchanged.  `hip(ml)` uploads it once and returns a `MultiLevel` whose workspace type parameter TW is
# `HipWorkspace`.  ONE method specialised on that type forwards `_solve!` to the C ABI, so `_solve(ml, b)`,
# `aspreconditioner(ml)`, `ldiv!`, `\`, `solve(A, b, RugeStubenAMG())` (CommonSolve `init` / `solve!`) keep
# their exact signatures (multilevel.jl:152-198,252-264, preconditioner.jl:10-24).  The per-level hooks of
# the reference (`smooth!(x, s, b)`, `mul!(y, A, x)`) dispatch on smoother caches and matrix types that are
# NOT replaced here (the levels keep their host matrices); their device counterparts are exposed as plain
# functions `level_smooth!` / `level_mul!` instead of new `smooth!` / `mul!` methods.
module AMGHip

using AlgebraicMultigrid
using SparseArrays, LinearAlgebra
using Printf
import AlgebraicMultigrid: MultiLevel, Level, _solve!, Cycle, V, W, F, GaussSeidel, Jacobi, SOR,
                           ForwardSweep, BackwardSweep, SymmetricSweep, Pinv, QRSolver,
                           HermitianSymmetry, NoSymmetry, FastGSSmoother, FastJacobiSmoother,
                           FastSORSmoother, aspreconditioner, ruge_stuben, smoothed_aggregation

const libamghip = get(ENV, "LIBAMGHIP", "libamghip.so")
# the Float32 instance: the same source compiled with amgh_real = float (include/amghip.h), same entry points
const libamghip_f32 = get(ENV, "LIBAMGHIP_F32", "libamghip_f32.so")

struct AMGHipError <: Exception
    rc::Cint
end
Base.showerror(io::IO, e::AMGHipError) =
    print(io, "libamghip: ", unsafe_string(ccall((:amgh_strerror, libamghip), Cstring, (Cint,), e.rc)))
check(rc) = rc == 0 ? nothing : throw(AMGHipError(rc))

# amgh_smoother_t (include/amghip.h); the relaxation factor stays Float64 in both instances
struct CSmoother
    kind::Int32; sweep::Int32; iter::Int32; pad::Int32; omega::Float64
end
sweepcode(::ForwardSweep) = Int32(0); sweepcode(::BackwardSweep) = Int32(1); sweepcode(::SymmetricSweep) = Int32(2)
csmoother(s::FastGSSmoother{S}) where {S} = CSmoother(1, sweepcode(S()), s.iter, 0, 1.0)
csmoother(s::FastJacobiSmoother) = CSmoother(2, 2, s.iter, 0, Float64(s.ω))
csmoother(s::FastSORSmoother{S}) where {S} = CSmoother(3, sweepcode(S()), s.iter, 0, Float64(s.ω))
# NoSymmetry caches map to the same kernels on the true rows (S == A); see amgh_push_level.

"""
Workspace type that marks a MultiLevel as resident on the GPU (replaces MultiLevelWorkspace{TX,bs}); `T` is the
arithmetic type of the handle = the instance of the library it lives in (Float64: libamghip, Float32: libamghip_f32).
`keep` holds everything the library may call back into or read later (the coarse-solver closure and its
`@cfunction` trampoline): they must live as long as the handle.
"""
mutable struct HipWorkspace{T}
    handle::Ptr{Cvoid}
    bs::Int
    keep::Vector{Any}
    function HipWorkspace{T}(h, bs, keep) where {T}
        w = new{T}(h, bs, keep)
        finalizer(destroy!, w)
    end
end
Base.eltype(::HipWorkspace{T}) where {T} = T
destroy!(w::HipWorkspace{Float64}) = ccall((:amgh_destroy, libamghip), Cvoid, (Ptr{Cvoid},), w.handle)
destroy!(w::HipWorkspace{Float32}) = ccall((:amgh_destroy, libamghip_f32), Cvoid, (Ptr{Cvoid},), w.handle)

# 0-based int32 CSR arrays of M given Julia's 1-based CSC of M' (CSC arrays of X are CSR arrays of X')
csr_of_transpose(::Type{T}, X::SparseMatrixCSC) where {T} = (Int32.(X.colptr .- 1), Int32.(X.rowval .- 1), T.(X.nzval))
csr(::Type{T}, X::SparseMatrixCSC) where {T} = csr_of_transpose(T, copy(X'))
csr(::Type{T}, X::Adjoint{<:Any,<:SparseMatrixCSC}) where {T} = csr_of_transpose(T, parent(X))   # lazy adjoint: already the CSR

cyclecode(::V) = Cint(0); cyclecode(::W) = Cint(1); cyclecode(::F) = Cint(2)

const HipML{T} = MultiLevel{<:Any,<:Any,<:Any,<:Any,<:Any,<:Any,HipWorkspace{T}}

"""
    hip(ml::MultiLevel; device = 0, bs = 1) -> MultiLevel

Upload the hierarchy to HBM (amgh_create / amgh_push_level / amgh_set_coarse / amgh_finalize).
`bs` = workspace block size (the reference's `Val{bs}`, multilevel.jl:28-35): `_solve!` then takes n x bs blocks.
The arithmetic type is eltype of the hierarchy (Float64 or Float32; anything else is promoted to Float64).
"""
function hip(ml::MultiLevel; kwargs...)
    A = isempty(ml.levels) ? ml.final_A : ml.levels[1].A
    hip(eltype(A) === Float32 ? Float32 : Float64, ml; kwargs...)
end

# one set of methods per instance of the library: `ccall` needs the library as a constant
for (T, lib) in ((Float64, :libamghip), (Float32, :libamghip_f32))
@eval begin

function hip(::Type{$T}, ml::MultiLevel; device::Integer = 0, bs::Integer = 1, symmetry = HermitianSymmetry())
    h = Ref{Ptr{Cvoid}}(C_NULL)
    keep = Any[]
    check(ccall((:amgh_create, $lib), Cint, (Ref{Ptr{Cvoid}}, Cint, Cint), h, device, bs))
    for lev in ml.levels
        A = lev.A
        n, nc = size(lev.P)
        Ar, Ac, Av = csr($T, A)                               # true rows: mul!(res, A, x)
        sym = symmetry isa HermitianSymmetry
        # the "fast" smoothers read CSC column i as row i (smoother.jl:81-86): S = CSC arrays as CSR
        S = (sym && !issymmetric(A)) ? csr_of_transpose($T, A) : nothing
        Pr, Pc, Pv = csr($T, lev.P); Rr, Rc, Rv = csr($T, lev.R)
        pre, post = Ref(csmoother(lev.presmoother)), Ref(csmoother(lev.postsmoother))
        GC.@preserve Ar Ac Av S Pr Pc Pv Rr Rc Rv check(ccall((:amgh_push_level, $lib), Cint,
            (Ptr{Cvoid}, Int64, Int64, Ptr{Int32}, Ptr{Int32}, Ptr{$T}, Ptr{Int32}, Ptr{Int32}, Ptr{$T},
             Ptr{Int32}, Ptr{Int32}, Ptr{$T}, Ptr{Int32}, Ptr{Int32}, Ptr{$T}, Ref{CSmoother}, Ref{CSmoother}),
            h[], n, nc, Ar, Ac, Av,
            S === nothing ? C_NULL : S[1], S === nothing ? C_NULL : S[2], S === nothing ? C_NULL : S[3],
            Pr, Pc, Pv, Rr, Rc, Rv, pre, post))
    end
    fA = ml.final_A
    n = size(fA, 1)
    fr, fc, fv = csr($T, fA)
    cs = ml.coarse_solver
    if n <= 2048
        # Pinv: the stored pinv(Matrix(A)) (coarse_solver.jl:11).  QRSolver (coarse_solver.jl:66-81) solves with a
        # pivoted QR, i.e. least squares when final_A is singular: its dense stand-in is the pseudo-inverse too,
        # NOT inv(), which throws on a singular coarse matrix (e.g. the pure-Neumann Poisson problem).
        op = cs isa Pinv ? Matrix{$T}(cs.pinvA) : Matrix{$T}(pinv(Matrix(fA)))
        check(ccall((:amgh_set_coarse, $lib), Cint,
                    (Ptr{Cvoid}, Int64, Ptr{Int32}, Ptr{Int32}, Ptr{$T}, Ptr{$T}), h[], n, fr, fc, fv, op))
    else
        # pluggable host coarse solver: the reference's `(cs)(x, b)` protocol through a C callback.  The closure
        # and its trampoline are stored in the workspace: the library calls them on every coarse solve.
        closure = (user, b, x, n) -> begin
            cs(unsafe_wrap(Array, x, n), unsafe_wrap(Array, b, n)); Cint(0)
        end
        cb = @cfunction($(Expr(:$, :closure)), Cint, (Ptr{Cvoid}, Ptr{$T}, Ptr{$T}, Int64))
        push!(keep, closure, cb, cs)
        check(ccall((:amgh_set_coarse_host, $lib), Cint,
                    (Ptr{Cvoid}, Int64, Ptr{Int32}, Ptr{Int32}, Ptr{$T}, Ptr{Cvoid}, Ptr{Cvoid}),
                    h[], n, fr, fc, fv, cb, C_NULL))
    end
    check(ccall((:amgh_finalize, $lib), Cint, (Ptr{Cvoid},), h[]))
    # the collapsed coarse tail's dense operator for V-cycles now (W / F: at their first cycle); multilevel.jl:227-231
    check(ccall((:amgh_tail_dense_build, $lib), Cint, (Ptr{Cvoid}, Cint), h[], 0))
    MultiLevel(ml.levels, ml.final_A, ml.coarse_solver, ml.presmoother, ml.postsmoother,
               HipWorkspace{$T}(h[], Int(bs), keep))
end

# _solve!(x, ml, b, cycle; maxiter, abstol, reltol, verbose, log, calculate_residual)  multilevel.jl:158-198
# b, x: vectors (bs = 1) or n x bs matrices — Julia's column-major layout is what amgh_solve takes.
# Any other eltype of x / b is converted to the handle's (promote_type as multilevel.jl:154 does, then the handle's).
function AlgebraicMultigrid._solve!(x::AbstractVecOrMat, ml::HipML{$T}, b::AbstractVecOrMat,
                                    cycle::Cycle = V();
                                    maxiter::Int = 100, abstol::Real = 0.0, reltol::Real = sqrt(eps(real(eltype(b)))),
                                    verbose::Bool = false, log::Bool = false, calculate_residual = true, kwargs...)
    size(b, 2) == ml.workspace.bs ||
        throw(DimensionMismatch("hierarchy was uploaded with block size $(ml.workspace.bs), b has $(size(b, 2)) columns"))
    size(x) == size(b) || throw(DimensionMismatch("x and b differ in size"))
    xs, bsd = Array{$T}(x), Array{$T}(b)     # dense, contiguous
    hist = zeros($T, maxiter + 1)
    iters = Ref{Cint}(0)
    check(ccall((:amgh_solve, $lib), Cint,
                (Ptr{Cvoid}, Ptr{$T}, Ptr{$T}, Cint, Cint, Float64, Float64, Cint, Ptr{$T}, Ref{Cint}),
                ml.workspace.handle, bsd, xs, cyclecode(cycle), maxiter, abstol, reltol, calculate_residual, hist, iters))
    copyto!(x, xs)
    if verbose && calculate_residual
        for i in 1:iters[]
            @printf "Norm of residual at iteration %6d is %.4e\n" i hist[i + 1]
        end
    end
    log ? (x, hist[1:(calculate_residual ? iters[] + 1 : 1)]) : x
end

# Per-level hooks on the device (the counterparts of `smooth!(x, levels[l].presmoother, b)` and `mul!(y, op, x)` on
# level `level`, 1-based); A, P, R of the level by `which`.
function level_smooth!(x::Vector{$T}, ml::HipML{$T}, level::Integer, b::Vector{$T}; post::Bool = false)
    check(ccall((:amgh_level_smooth, $lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{$T}, Ptr{$T}),
                ml.workspace.handle, level - 1, post, x, b))
    x
end
function level_mul!(y::Vector{$T}, ml::HipML{$T}, level::Integer, which::Symbol, x::Vector{$T})
    code = which === :A ? 0 : which === :P ? 1 : which === :R ? 2 : throw(ArgumentError("which must be :A, :P or :R"))
    check(ccall((:amgh_level_spmv, $lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{$T}, Ptr{$T}),
                ml.workspace.handle, level - 1, code, x, y))
    y
end

# device-resident preconditioned CG: IterativeSolvers' cg(A, b; Pl = aspreconditioner(ml)) in one call
function cg(ml::HipML{$T}, b::Vector{$T}; cycle::Cycle = V(), maxiter::Int = length(b), abstol = 0.0,
            reltol = sqrt(eps($T)))
    x = zeros($T, length(b)); hist = zeros($T, maxiter + 1); iters = Ref{Cint}(0)
    check(ccall((:amgh_pcg, $lib), Cint,
                (Ptr{Cvoid}, Ptr{$T}, Ptr{$T}, Cint, Cint, Cint, Float64, Float64, Ptr{$T}, Ref{Cint}),
                ml.workspace.handle, b, x, cyclecode(cycle), 1, maxiter, abstol, reltol, hist, iters))
    x, hist[1:iters[] + 1]
end

end # @eval
end # for (T, lib)

# `aspreconditioner(ml)`, `ldiv!`, `\` need no new methods: preconditioner.jl:12-19 calls `_solve!` with
# maxiter = 1, calculate_residual = false, which lands in the method above.  CommonSolve's
# `solve(A, b, RugeStubenAMG())` = `init` + `solve!` (multilevel.jl:252-264) runs on the GPU once `init` wraps
# the hierarchy:
struct HipRugeStubenAMG <: AlgebraicMultigrid.AMGAlg end
struct HipSmoothedAggregationAMG <: AlgebraicMultigrid.AMGAlg end
AlgebraicMultigrid.init(::HipRugeStubenAMG, A, b, args...; kwargs...) =
    AlgebraicMultigrid.AMGSolver(hip(ruge_stuben(A; kwargs...)), b)
AlgebraicMultigrid.init(::HipSmoothedAggregationAMG, A, b; kwargs...) =
    AlgebraicMultigrid.AMGSolver(hip(smoothed_aggregation(A; kwargs...)), b)

# LinearSolve `precs` builders (precs.jl:7-38): `(builder)(A, p) -> (Pl, I)`
struct HipRugeStubenPreconBuilder{Tk}
    blocksize::Int
    kwargs::Tk
end
HipRugeStubenPreconBuilder(; blocksize = 1, kwargs...) = HipRugeStubenPreconBuilder(blocksize, kwargs)
(b::HipRugeStubenPreconBuilder)(A::SparseArrays.AbstractSparseMatrixCSC, p) =
    (aspreconditioner(hip(ruge_stuben(SparseMatrixCSC(A), Val{b.blocksize}; b.kwargs...); bs = b.blocksize)), I)
struct HipSmoothedAggregationPreconBuilder{Tk}
    blocksize::Int
    kwargs::Tk
end
HipSmoothedAggregationPreconBuilder(; blocksize = 1, kwargs...) = HipSmoothedAggregationPreconBuilder(blocksize, kwargs)
(b::HipSmoothedAggregationPreconBuilder)(A::SparseArrays.AbstractSparseMatrixCSC, p) =
    (aspreconditioner(hip(smoothed_aggregation(SparseMatrixCSC(A), Val{b.blocksize}; b.kwargs...); bs = b.blocksize)), I)

# ---- row-sharded hierarchy over N GPUs (amgh_dist_*, include/amghip.h; Float64 instance only) ------------------
# One Julia process per GPU (e.g. under MPI.jl).  Rank 0 makes the RCCL id, the host broadcasts its 128 bytes
# (`MPI.Bcast!`), every rank pushes ITS rows of the sharded levels with global column indices, rank 0 passes
# the collapsed levels as an ordinary uploaded hierarchy.
rccl_unique_id() = (id = zeros(UInt8, 128); check(ccall((:amgh_dist_unique_id, libamghip), Cint, (Ptr{UInt8},), id)); id)

mutable struct HipSharded
    handle::Ptr{Cvoid}
    tail::Union{Nothing,MultiLevel}     # keeps the collapsed levels' handle alive
    r0::Int; r1::Int                    # this rank's fine rows (0-based, half open)
end

rowcuts(n, N) = Int64[div(p * n, N) for p in 0:N]
function localrows(M::SparseMatrixCSC, r0, r1)      # rows [r0, r1) of M' as CSR with global columns (= columns of M)
    lo, hi = M.colptr[r0 + 1], M.colptr[r1 + 1] - 1
    (Int32.(M.colptr[r0 + 1:r1 + 1] .- lo), Int32.(M.rowval[lo:hi] .- 1), Float64.(M.nzval[lo:hi]))
end

"""
    hip_sharded(ml, rank, nranks, id; device = rank, shard_min_rows = 200_000, gs_exact = true) -> HipSharded

Levels with at least `shard_min_rows` rows are partitioned by contiguous row ranges; the rest is uploaded on rank 0
(`hip`) and handed over with amgh_dist_set_tail.  Assumes symmetric level operators (A' = A), as the sharded
Python driver does.

`id::Vector{UInt8}` (the 128 bytes of `amgh_dist_unique_id`, broadcast from rank 0) selects the RCCL transport;
`id::AbstractString` (a fresh shared-memory name in `shm_open` syntax, e.g. "/amgh_1234", the same on every rank)
selects the IPC transport: one process per rank, hipIpc peer-mapped send buffers, hand-off by flags the streams write
and wait on — ranks may then share a GPU, which RCCL refuses.  `gs_exact = true`: Gauss-Seidel / SOR sweep the whole
level in lexicographic order (the ranks in turn: the reference's iterate); `false`: every shard at once, halo frozen
per directional sweep (`amgh_dist_set_gs_mode`).
"""
function hip_sharded(ml::MultiLevel, rank::Integer, nranks::Integer, id::Union{Vector{UInt8},AbstractString};
                     device::Integer = rank, shard_min_rows::Integer = 200_000, gs_exact::Bool = true)
    d = Ref{Ptr{Cvoid}}(C_NULL)
    if id isa AbstractString
        check(ccall((:amgh_dist_create_ipc, libamghip), Cint, (Ref{Ptr{Cvoid}}, Cint, Cint, Cint, Cstring),
                    d, device, rank, nranks, id))
    else
        check(ccall((:amgh_dist_create_rccl, libamghip), Cint, (Ref{Ptr{Cvoid}}, Cint, Cint, Cint, Ptr{UInt8}),
                    d, device, rank, nranks, id))
    end
    sizes = [[size(l.A, 1) for l in ml.levels]; size(ml.final_A, 1)]
    lc = 0
    while lc < length(ml.levels) && sizes[lc + 1] >= shard_min_rows && sizes[lc + 1] >= 8 * nranks
        lc += 1
    end
    for l in 1:lc
        lev = ml.levels[l]
        n, nc = sizes[l], sizes[l + 1]
        cuts = rowcuts(n, nranks)
        ccuts = l < lc ? rowcuts(nc, nranks) : Int64[0; fill(nc, nranks)]
        r0, r1, c0, c1 = cuts[rank + 1], cuts[rank + 2], ccuts[rank + 1], ccuts[rank + 2]
        Ar, Ac, Av = localrows(lev.A, r0, r1)                          # A symmetric: CSC columns are the rows
        Pr, Pc, Pv = localrows(SparseMatrixCSC(lev.P'), r0, r1)        # rows of P = columns of P'
        Rr, Rc, Rv = localrows(SparseMatrixCSC(lev.R'), c0, c1)
        pre, post = Ref(csmoother(lev.presmoother)), Ref(csmoother(lev.postsmoother))
        GC.@preserve cuts ccuts Ar Ac Av Pr Pc Pv Rr Rc Rv check(ccall((:amgh_dist_push_level, libamghip), Cint,
            (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64},
             Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64},
             Ref{CSmoother}, Ref{CSmoother}),
            d[], n, nc, cuts, ccuts, Ar, Ac, Av, C_NULL, C_NULL, C_NULL, Pr, Pc, Pv, Rr, Rc, Rv, pre, post))
    end
    tail = nothing
    if rank == 0
        tail = hip(Float64, MultiLevel(ml.levels[lc + 1:end], ml.final_A, ml.coarse_solver, ml.presmoother, ml.postsmoother,
                              ml.workspace); device = device)
        check(ccall((:amgh_dist_set_tail, libamghip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), d[], tail.workspace.handle))
    end
    check(ccall((:amgh_dist_finalize, libamghip), Cint, (Ptr{Cvoid},), d[]))
    # Gauss-Seidel / SOR across shards: exact lexicographic order (the reference's iterate; default) or the frozen-halo hybrid
    gs_exact || check(ccall((:amgh_dist_set_gs_mode, libamghip), Cint, (Ptr{Cvoid}, Cint), d[], 0))
    cuts0 = rowcuts(sizes[1], nranks)
    s = HipSharded(d[], tail, lc > 0 ? cuts0[rank + 1] : (rank == 0 ? 0 : sizes[1]), lc > 0 ? cuts0[rank + 2] : sizes[1])
    finalizer(s -> ccall((:amgh_dist_destroy, libamghip), Cvoid, (Ptr{Cvoid},), s.handle), s)
end

# device pointers in (this rank's rows), one V / W / F cycle with x = 0 (ldiv!)
precond_apply_d!(s::HipSharded, r_d::Ptr{Float64}, z_d::Ptr{Float64}, cycle::Cycle = V()) =
    check(ccall((:amgh_dist_precond_apply_d, libamghip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Cint),
                s.handle, r_d, z_d, cyclecode(cycle)))

end # module
