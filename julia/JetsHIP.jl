# JetsHIP.jl -- Julia binding of libjetship.so (include/jetship.h) for Jets.jl.
#
# STATUS: WRITTEN, NOT EXECUTED.  This image has no Julia toolchain, so this file has never been
# parsed or run; it documents, in the reference's own language, the `ccall` stubs a Jets.jl
# maintainer adds to put JopBlock mul! on an MI355X.  The same ABI is exercised end to end by the
# Python binding (jets.jl_amd/_ffi.py) and the GPU test-suite.
#
# Everything is additive: new array types (`HipArray`, `HipBlockArray`) plus methods of existing
# Jets.jl generics that dispatch on them.  No reference source is modified.
module JetsHIP

using Jets, LinearAlgebra
import Jets: JetBSpace, JetSpace, JetAbstractSpace, Jop, JopLn, JopAdjoint, Jet, jet, state, domain,
             getblock, getblock!, setblock!, indices, nblocks, space, JopZeroBlock_df!, JetBlock_df!, JetBlock_df′!

const LIB = get(ENV, "JETSHIP_LIB", "libjetship.so")

# ---------------------------------------------------------------- errors (src/Jets.jl:131,179,1116: plain error(...))
check(status::Cint) = status == 0 ? nothing : error("libjetship: " * unsafe_string(ccall((:jh_last_error, LIB), Cstring, ())))

init(device::Integer=0) = check(ccall((:jh_init, LIB), Cint, (Cint,), device))
synchronize() = check(ccall((:jh_synchronize, LIB), Cint, ()))

dtype_code(::Type{Float32}) = Cint(0)
dtype_code(::Type{Float64}) = Cint(1)
dtype_code(::Type{ComplexF32}) = Cint(2)
dtype_code(::Type{ComplexF64}) = Cint(3)

# ---------------------------------------------------------------- device vectors
# One HBM slab; block i at element offset R.indices[i][1]-1 (src/Jets.jl:742-748).
mutable struct HipBlockArray{T} <: AbstractArray{T,1}
    handle::Ptr{Cvoid}
    spaces::Vector{<:JetAbstractSpace}
    indices::Vector{UnitRange{Int}}
    parent::Any                       # keeps the owner alive for views
    function HipBlockArray{T}(handle, spaces, indices, parent=nothing) where {T}
        x = new{T}(handle, spaces, indices, parent)
        finalizer(x -> ccall((:jh_bvec_destroy, LIB), Cint, (Ptr{Cvoid},), x.handle), x)
        x
    end
end

# a plain N-d array is a one-block slab (domain of a one-column block operator, src/Jets.jl:927)
mutable struct HipArray{T,N} <: AbstractArray{T,N}
    handle::Ptr{Cvoid}
    dims::NTuple{N,Int}
    parent::Any
    function HipArray{T,N}(handle, dims, parent=nothing) where {T,N}
        x = new{T,N}(handle, dims, parent)
        finalizer(x -> ccall((:jh_bvec_destroy, LIB), Cint, (Ptr{Cvoid},), x.handle), x)
        x
    end
end

Base.size(x::HipBlockArray) = (x.indices[end][end],)                      # src/Jets.jl:818
Base.size(x::HipArray) = x.dims
Jets.nblocks(x::HipBlockArray) = length(x.indices)                       # :860
Jets.indices(x::HipBlockArray, i) = x.indices[i]                         # :858
Jets.space(x::HipBlockArray) = JetBSpace(x.spaces)                       # :814

function _create(lens::Vector{Int}, ::Type{T}) where {T}
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:jh_bvec_create, LIB), Cint, (Int64, Ptr{Int64}, Cint, Ref{Ptr{Cvoid}}), length(lens), lens, dtype_code(T), h))
    h[]
end

# zeros(R) / Array(R) on the device (src/Jets.jl:105-108, 922-924); device storage is always zero-filled
hipzeros(R::JetBSpace{T}) where {T} = HipBlockArray{T}(_create([length(R.indices[i]) for i = 1:length(R.indices)], T), R.spaces, R.indices)
hipzeros(R::JetSpace{T,N}) where {T,N} = HipArray{T,N}(_create([length(R)], T), size(R))
hiprand(R::JetAbstractSpace; seed=1, stream=0) = (x = hipzeros(R); check(ccall((:jh_fill_uniform, LIB), Cint, (Ptr{Cvoid}, UInt64, UInt64, Int64), x.handle, seed, stream, 0)); x)

# getblock(x, i): by reference (src/Jets.jl:914) -- 1-based i -> 0-based block
function Jets.getblock(x::HipBlockArray{T}, iblock) where {T}
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:jh_bvec_view, LIB), Cint, (Ptr{Cvoid}, Int64, Int64, Ref{Ptr{Cvoid}}), x.handle, iblock - 1, 1, h))
    HipArray{T,ndims(x.spaces[iblock])}(h[], size(x.spaces[iblock]), x)
end
Jets.getblock(x::HipArray, iblock) = x                                    # :918
# getblock!(x, i, xblock) (:915) and setblock!(x, i, v) (:916) for host arrays and scalars
Jets.getblock!(x::HipBlockArray, iblock, out::Array) = (check(ccall((:jh_getblock_copy, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Cint), x.handle, iblock - 1, out, 0)); out)
Jets.setblock!(x::HipBlockArray, iblock, v::Array) = check(ccall((:jh_setblock_copy, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Cint), x.handle, iblock - 1, v, 0))
Jets.setblock!(x::HipBlockArray, iblock, a::Number) = check(ccall((:jh_setblock_fill, LIB), Cint, (Ptr{Cvoid}, Int64, Cdouble, Cdouble), x.handle, iblock - 1, real(a), imag(a)))

# convert(Array, x) (:862-868)
function Base.convert(::Type{Array}, x::Union{HipBlockArray{T},HipArray{T}}) where {T}
    out = Vector{T}(undef, length(x))
    check(ccall((:jh_download, LIB), Cint, (Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}), x.handle, 0, length(x), out))
    x isa HipArray ? reshape(out, size(x)) : out
end

# a page-locked host array (jh_host_alloc) for the copies above: no first-touch page faults under the DMA
function pinned_array(::Type{T}, dims::Integer...) where {T}
    p = Ref{Ptr{Cvoid}}()
    check(ccall((:jh_host_alloc, LIB), Cint, (Csize_t, Ref{Ptr{Cvoid}}), prod(dims) * sizeof(T), p))
    a = unsafe_wrap(Array, convert(Ptr{T}, p[]), dims; own=false)
    finalizer(_ -> ccall((:jh_host_free, LIB), Cint, (Ptr{Cvoid},), p[]), a)
    a
end

# fill!, norm, dot, extrema (:834-885)
Base.fill!(x::Union{HipBlockArray,HipArray}, a) = (check(ccall((:jh_fill, LIB), Cint, (Ptr{Cvoid}, Cdouble, Cdouble), x.handle, real(a), imag(a))); x)
function LinearAlgebra.norm(x::Union{HipBlockArray{T},HipArray{T}}, p::Real=2) where {T}
    out = Ref{Cdouble}()
    check(ccall((:jh_norm, LIB), Cint, (Ptr{Cvoid}, Cdouble, Ref{Cdouble}), x.handle, p, out))
    float(real(T))(out[])
end
function LinearAlgebra.dot(x::Union{HipBlockArray{T},HipArray{T}}, y::Union{HipBlockArray{T},HipArray{T}}) where {T}
    re, im = Ref{Cdouble}(), Ref{Cdouble}()
    check(ccall((:jh_dot, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Cdouble}, Ref{Cdouble}), x.handle, y.handle, re, im))
    T <: Complex ? T(re[], im[]) : T(re[])
end
function Base.extrema(x::Union{HipBlockArray{T},HipArray{T}}) where {T<:Real}
    mn, mx = Ref{Cdouble}(), Ref{Cdouble}()
    check(ccall((:jh_extrema, LIB), Cint, (Ptr{Cvoid}, Ref{Cdouble}, Ref{Cdouble}), x.handle, mn, mx))
    T(mn[]), T(mx[])
end

# y .= c1*x1 .+ c2*x2 .+ ...   (the BlockArrayStyle copyto!, :905-911, for linear combinations)
function lincomb!(y, coefs::Vector{<:Number}, xs::Vector)
    cf = Float64[]
    for c in coefs
        push!(cf, real(c), imag(c))
    end
    hs = Ptr{Cvoid}[x.handle for x in xs]
    check(ccall((:jh_lincomb, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Ptr{Cvoid}}), y.handle, length(xs), cf, hs))
    y
end

# ---------------------------------------------------------------- broadcast: any elementwise expression, one fused pass
# copyto!(dest::BlockArray, bc::Broadcasted{BlockArrayStyle}) (src/Jets.jl:899-911) for device vectors: print the
# Broadcasted tree as a C expression over x0.. (vector leaves) and s0.. (scalar leaves), let libjetship compile it once
# with hiprtc (jh_bcast_compile) and stream the slabs in ONE kernel.  `a*u .+ b*v .+ c*w`, `exp.(-u.^2) .* v`, ...
const _cfun = Dict{Any,String}(+ => "+", - => "-", * => "*", / => "/", exp => "exp", log => "log", sqrt => "sqrt", sin => "sin",
                               cos => "cos", tanh => "tanh", abs => "abs", abs2 => "abs2", conj => "conj", real => "real",
                               imag => "imag", sign => "sign", max => "fmax", min => "fmin")
_emit(x::Union{HipBlockArray,HipArray}, vecs, scals) = (i = findfirst(v -> v === x, vecs); i === nothing && (push!(vecs, x); i = length(vecs)); "x$(i-1)")
_emit(x::Number, vecs, scals) = (push!(scals, x); "s$(length(scals)-1)")
_emit(x::Base.RefValue, vecs, scals) = _emit(x[], vecs, scals)
function _emit(bc::Base.Broadcast.Broadcasted, vecs, scals)
    if bc.f === Base.literal_pow                                  # u.^p with a literal p: args = (Ref(^), u, Ref(Val(p)))
        p = typeof(bc.args[3][]).parameters[1]
        base = _emit(bc.args[2], vecs, scals)
        return (p isa Integer && 1 <= p <= 4) ? "(" * join(fill(base, p), " * ") * ")" : "pow($base, $p)"   # u.^2 -> u*u, like Julia
    end
    f, args = bc.f, map(a -> _emit(a, vecs, scals), bc.args)
    op = get(_cfun, f, nothing)
    op === nothing && error("broadcast of $(f) over device vectors is not supported")
    op in ("+", "-", "*", "/") ? (length(args) == 1 ? "($op$(args[1]))" : "(" * join(args, " $op ") * ")") : "$op(" * join(args, ", ") * ")"
end
const _bcast_programs = Dict{Tuple{String,DataType,Int,Int},Ptr{Cvoid}}()
function Base.copyto!(dest::Union{HipBlockArray{T},HipArray{T}}, bc::Base.Broadcast.Broadcasted) where {T}
    vecs, scals = Any[], Number[]
    expr = _emit(Base.Broadcast.flatten(bc), vecs, scals)
    prog = get!(_bcast_programs, (expr, T, length(vecs), length(scals))) do
        h = Ref{Ptr{Cvoid}}()
        check(ccall((:jh_bcast_compile, LIB), Cint, (Cstring, Cint, Cint, Cint, Ref{Ptr{Cvoid}}), expr, dtype_code(T), length(vecs), length(scals), h))
        h[]
    end
    sc = Cdouble[]
    foreach(a -> (push!(sc, real(a)); push!(sc, imag(a))), scals)
    check(ccall((:jh_bcast_apply, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Ptr{Cdouble}), prog, dest.handle, Ptr{Cvoid}[v.handle for v in vecs], sc))
    dest
end

# ---------------------------------------------------------------- device-native operator kinds
# recognised by typeof(df!) exactly as iszero/isblockop do (src/Jets.jl:949, 1097)
JopHipDiagonal_df!(d, m; diagonal, kwargs...) = (check(ccall((:jh_hadamard, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint), d.handle, diagonal.handle, m.handle, 0)); d)
JopHipDiagonal_df′!(m, d; diagonal, kwargs...) = (check(ccall((:jh_hadamard, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint), m.handle, diagonal.handle, d.handle, 1)); m)
function JopHipDiagonal(diag::HipArray{T,N}) where {T,N}
    spc = JetSpace(T, size(diag))
    JopLn(;df! = JopHipDiagonal_df!, df′! = JopHipDiagonal_df′!, dom = spc, rng = spc, s = (diagonal=diag,))
end

# the reference's nonlinear fixture JopBar (test/runtests.jl:19-24) on device vectors: kind SQUARE
JopHipSquare_f!(d, m; kwargs...) = (check(ccall((:jh_hadamard, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint), d.handle, m.handle, m.handle, 0)); d)
JopHipSquare_df!(δd, δm; mₒ, kwargs...) = (check(ccall((:jh_hadamard, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint), δd.handle, mₒ.handle, δm.handle, 2)); δd)
JopHipSquare_df′!(δm, δd; mₒ, kwargs...) = (check(ccall((:jh_hadamard, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint), δm.handle, mₒ.handle, δd.handle, 3)); δm)
JopHipSquare(spc::JetSpace) = JopNl(f! = JopHipSquare_f!, df! = JopHipSquare_df!, df′! = JopHipSquare_df′!, dom = spc, rng = spc)

struct jh_block_desc          # mirrors include/jetship.h
    kind::Int32
    adjoint::Int32
    coeff::Ptr{Cvoid}
    scale_re::Cdouble
    scale_im::Cdouble
    nr::Int64
    nc::Int64
end

function block_desc(op::Jop)
    adj = op isa JopAdjoint
    base = adj ? op.op : op
    j = jet(base)
    nr, nc = length(range(base)), length(domain(base))
    if op isa JopNl
        return j.f! === JopHipSquare_f! ? jh_block_desc(5, 0, C_NULL, 0, 0, nr, nc) : nothing
    elseif j.df! === JopZeroBlock_df!
        return jh_block_desc(0, adj, C_NULL, 0, 0, nr, nc)
    elseif j.df! === JopHipDiagonal_df!
        return jh_block_desc(3, adj, _device_ptr(state(base).diagonal), 0, 0, nr, nc)
    elseif j.df! === Jets._constdiag_df!
        a = state(base).a
        return jh_block_desc(2, adj, C_NULL, real(a), imag(a), nr, nc)
    end
    nothing            # not device-native: the reference's per-block loop handles it
end

function _device_ptr(x)
    p = Ref{Ptr{Cvoid}}()
    check(ccall((:jh_bvec_info, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Cint}, Ref{Ptr{Cvoid}}), x.handle, C_NULL, C_NULL, C_NULL, p))
    p[]
end

const _handles = IdDict{Any,Ptr{Cvoid}}()        # ops matrix -> jh_blockop*

function native_handle(ops::AbstractMatrix{<:Jop}, ::Type{T}) where {T}
    get!(_handles, ops) do
        descs = [block_desc(ops[i,j]) for i = 1:size(ops,1), j = 1:size(ops,2)]      # column-major == the C layout
        any(isnothing, descs) && return C_NULL
        row_len = Int64[length(range(ops[i,1])) for i = 1:size(ops,1)]
        col_len = Int64[length(domain(ops[1,j])) for j = 1:size(ops,2)]
        h = Ref{Ptr{Cvoid}}()
        check(ccall((:jh_blockop_create, LIB), Cint, (Int64, Int64, Ptr{jh_block_desc}, Ptr{Int64}, Ptr{Int64}, Cint, Ref{Ptr{Cvoid}}),
                    size(ops,1), size(ops,2), vec(convert(Matrix{jh_block_desc}, descs)), row_len, col_len, dtype_code(T), h))
        h[]
    end
end

# ---------------------------------------------------------------- the hot path: ONE ccall per mul!
# More specific methods of the reference's block loops (src/Jets.jl:1010-1057) for device vectors.
function Jets.JetBlock_df!(d::HipBlockArray{T}, m::Union{HipArray{T},HipBlockArray{T}}; ops, dom, rng, kwargs...) where {T}
    h = native_handle(ops, T)
    h == C_NULL && return invoke(Jets.JetBlock_df!, Tuple{AbstractArray,AbstractArray}, d, m; ops, dom, rng, kwargs...)
    check(ccall((:jh_blockop_mul, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), h, d.handle, m.handle))
    d
end

function Jets.JetBlock_df′!(m::Union{HipArray{T},HipBlockArray{T}}, d::HipBlockArray{T}; ops, dom, rng, kwargs...) where {T}
    h = native_handle(ops, T)
    h == C_NULL && return invoke(Jets.JetBlock_df′!, Tuple{AbstractArray,AbstractArray}, m, d; ops, dom, rng, kwargs...)
    check(ccall((:jh_blockop_mul_adj, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), h, m.handle, d.handle))
    m
end

# nonlinear block operators: JetBlock_f! (src/Jets.jl:988-1008) and the block point! (1059-1066).  The Jacobian then
# runs through the two methods above; they call `_point_native` first when the operator has nonlinear children.
function Jets.JetBlock_f!(d::HipBlockArray{T}, m::Union{HipArray{T},HipBlockArray{T}}; ops, dom, rng, kwargs...) where {T}
    h = native_handle(ops, T)
    h == C_NULL && return invoke(Jets.JetBlock_f!, Tuple{AbstractArray,AbstractArray}, d, m; ops, dom, rng, kwargs...)
    check(ccall((:jh_blockop_f, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), h, d.handle, m.handle))
    d
end

function Jets.point!(j::Jet{D,R,typeof(Jets.JetBlock_f!)}, mₒ::Union{HipArray{T},HipBlockArray{T}}) where {D,R,T}
    invoke(Jets.point!, Tuple{Jet{D,R,typeof(Jets.JetBlock_f!)},AbstractArray}, j, mₒ)   # children first (1062-1064)
    h = native_handle(state(j).ops, T)
    h == C_NULL || check(ccall((:jh_blockop_point, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), h, mₒ.handle))
    j
end

# (A' o A) * m fused (src/Jets.jl:530-534 over (A', A)): called from a JetComposite_df! method that
# recognises ops == (A', A) with A a native tall block operator
function normal_mul!(y::HipArray{T}, A::JopLn, m::HipArray{T}) where {T}
    h = native_handle(state(A).ops, T)
    check(ccall((:jh_blockop_normal_mul, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), h, y.handle, m.handle))
    y
end

# ---------------------------------------------------------------- solver halves and the multi-GPU exchange
# u <- alpha*(A v) + beta*u, returns ||u||   /   v <- alpha*(A' (in_scale*u)) + beta*v, returns ||v||   (LSQR / CGLS)
function mul_axpby!(u::HipBlockArray{T}, A::JopLn, v::HipArray{T}, alpha::Real, beta::Real) where {T}
    nrm2 = Ref{Cdouble}()
    check(ccall((:jh_blockop_mul_axpby, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Ref{Cdouble}),
                native_handle(state(A).ops, T), u.handle, v.handle, alpha, beta, nrm2))
    sqrt(nrm2[])
end
function mul_adj_axpby!(v::HipArray{T}, A::JopLn, u::HipBlockArray{T}, alpha::Real, beta::Real; in_scale::Real=1.0) where {T}
    nrm2 = Ref{Cdouble}()
    check(ccall((:jh_blockop_mul_adj_axpby, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Cdouble, Ref{Cdouble}),
                native_handle(state(A).ops, T), v.handle, u.handle, alpha, beta, in_scale, nrm2))
    sqrt(nrm2[])
end

# one whole Golub-Kahan step in one pass: u <- alpha*(A v) + beta*u ; w <- A'u ; returns ||u||   (3/5 of the bytes of the two halves)
function bidiag_step!(u::HipBlockArray{T}, w::HipArray{T}, A::JopLn, v::HipArray{T}, alpha::Real, beta::Real) where {T}
    nrm2 = Ref{Cdouble}()
    check(ccall((:jh_blockop_bidiag_step, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Ref{Cdouble}),
                native_handle(state(A).ops, T), u.handle, v.handle, w.handle, alpha, beta, nrm2))
    sqrt(nrm2[])
end

# the whole LSQR loop behind the ABI (the same call solves the row-partitioned problem after comm_init on every rank)
struct jh_lsqr_result; istop::Int32; itn::Int32; r1norm::Cdouble; r2norm::Cdouble; anorm::Cdouble; acond::Cdouble; arnorm::Cdouble; xnorm::Cdouble; end
function hip_lsqr!(x::HipArray{T}, A::JopLn, b::HipBlockArray{T}; x0::Bool=false, damp=0.0, atol=1e-6, btol=1e-6, conlim=1e8, maxiter=100) where {T}
    res = Ref{jh_lsqr_result}()
    hist = Vector{Cdouble}(undef, 2 * maxiter)
    check(ccall((:jh_lsqr_solve, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint, Ref{jh_lsqr_result}, Ptr{Cdouble}),
                native_handle(state(A).ops, T), b.handle, x.handle, x0, damp, atol, btol, conlim, maxiter, 0, res, hist))   # b is overwritten (it becomes u)
    x, res[], reshape(hist, 2, :)[:, 1:res[].itn]
end

# one process per GPU: rank 0 makes the id, the host (MPI.jl, sockets, a file) ships it, every rank joins
comm_unique_id() = (id = Vector{UInt8}(undef, 128); check(ccall((:jh_comm_unique_id, LIB), Cint, (Ptr{UInt8},), id)); id)
comm_init(id::Vector{UInt8}, nranks::Integer, rank::Integer) = check(ccall((:jh_comm_init_rank, LIB), Cint, (Ptr{UInt8}, Cint, Cint), id, nranks, rank))
allreduce_sum!(x::Union{HipArray,HipBlockArray}) = (check(ccall((:jh_comm_allreduce_sum, LIB), Cint, (Ptr{Cvoid},), x.handle)); x)
function allreduce_scalars!(vals::Vector{Float64}, op::Symbol=:sum)
    check(ccall((:jh_comm_allreduce_scalars, LIB), Cint, (Ptr{Cdouble}, Cint, Cint), vals, length(vals), op === :sum ? 0 : (op === :max ? 1 : 2)))
    vals
end
# adjoint of a row-partitioned tall operator: local ordered sum, then the in-place all-reduce (src/Jets.jl:1045-1053 summed over ranks)
function mul_adj_partitioned!(m::HipArray{T}, A::JopLn, d_local::HipBlockArray{T}) where {T}
    check(ccall((:jh_blockop_mul_adj, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), native_handle(state(A).ops, T), m.handle, d_local.handle))
    allreduce_sum!(m)
end

# close(A) releases the device operator (src/Jets.jl:1120-1124 cascade)
function release!(ops)
    h = pop!(_handles, ops, C_NULL)
    h == C_NULL || ccall((:jh_blockop_destroy, LIB), Cint, (Ptr{Cvoid},), h)
    nothing
end

end # module
