# JetsHIP.jl -- Julia binding of libjetship.so (include/jetship.h) for Jets.jl.
#
# STATUS: WRITTEN, NOT EXECUTED.  This image has no Julia toolchain, so this file has never been parsed or run.  What is
# checked mechanically: every `ccall` below against the prototypes of include/jetship.h (symbol, arity, argument classes,
# struct layouts -- tests/test_julia_binding_static.py).  The same ABI is exercised end to end by the Python binding
# (jets.jl_amd/_ffi.py) and the GPU test-suite.
#
# How it drops in.  Jets.jl allocates every vector through its SPACES (`zeros(range(A))` in `A*m`, src/Jets.jl:399; the
# temporaries of JetComposite / JetSum / JetBlock, 524-538, 639-655, 990-1043), so device storage enters through a space:
#
#     R = HipSpace(Float32, 256, 256, 256)          # instead of JetSpace(Float32, 256, 256, 256)
#     A = @blockop [JopHipDiagonal(rand(R)) for i = 1:1024, j = 1:1]
#     m = rand(domain(A)); d = A*m; mt = A'*d; x = lsqr(vec(A), vec(d))       # unchanged user code
#
# `zeros/ones/rand/randn/Array(::HipSpace)` return `HipArray`s (HIP device memory); the same factories on a
# `JetBSpace` of HipSpaces return the reference's own `BlockArray` (src/Jets.jl:809-812) whose blocks are consecutive VIEWS
# OF ONE HBM SLAB laid out like `JetBSpace.indices` (742-748).  Everything the reference does with a BlockArray keeps
# working (getblock by reference, space(x), broadcasting machinery), and the more specific methods below replace the
# per-block loops by ONE `ccall` on the slab: norm/dot/extrema/fill!/convert/broadcast, and -- the hot path --
# JetBlock_f!/df!/df′! (988-1057), the fused A'∘A (530-534) and the LSQR loop.
#
# Everything is additive: new types (`HipSpace`, `HipArray`) plus methods of existing generics.  No reference source is modified.
module JetsHIP

using Jets, LinearAlgebra
import Jets: JetAbstractSpace, JetBSpace, JetSpace, BlockArray, BlockArrayStyle, Jop, JopLn, JopNl, JopAdjoint, Jet, jet, state,
             domain, getblock, getblock!, setblock!, indices, nblocks, space, point!, JopZeroBlock_df!, JetBlock_f!,
             JetBlock_df!, JetBlock_df′!, JetComposite_f!, JetComposite_df!, JetComposite_df′!, JetSum_df!, JetSum_df′!, _constdiag_df!, _constdiag_df′!, JetBlock

export HipSpace, HipArray, JopHipDiagonal, JopHipSquare, JopHipDense, hip_lsqr!, blocknorms, blockdots

const LIB = get(ENV, "JETSHIP_LIB", "libjetship.so")

# ---------------------------------------------------------------- errors (src/Jets.jl:131,179,1116: plain error(...))
check(status::Cint) = status == 0 ? nothing : error("libjetship: " * unsafe_string(ccall((:jh_last_error, LIB), Cstring, ())))
const ABI_VERSION = 4       # JETSHIP_ABI_VERSION of include/jetship.h: struct layouts (jh_block_desc) are part of it
function __init__()
    v = ccall((:jh_abi_version, LIB), Cint, ())
    v == ABI_VERSION || error("libjetship speaks ABI version $v, this binding $ABI_VERSION: rebuild the library")
end

const _inited = Ref(false)
init(device::Integer=0) = (check(ccall((:jh_init, LIB), Cint, (Cint,), device)); _inited[] = true; nothing)
trim() = check(ccall((:jh_trim, LIB), Cint, ()))          # hand the slab cache's device memory back to the driver (before another library needs it)
_ensure_init() = _inited[] || init(parse(Int, get(ENV, "JETSHIP_DEVICE", "0")))      # the first allocation picks the device
synchronize() = check(ccall((:jh_synchronize, LIB), Cint, ()))
tune!(name::AbstractString, value::Integer) = check(ccall((:jh_tune_set, LIB), Cint, (Cstring, Int64), name, value))   # e.g. tune!("adj_split", 0)

const HipEltype = Union{Float32,Float64,ComplexF32,ComplexF64}
dtype_code(::Type{Float32}) = Cint(0)
dtype_code(::Type{Float64}) = Cint(1)
dtype_code(::Type{ComplexF32}) = Cint(2)
dtype_code(::Type{ComplexF64}) = Cint(3)

# ---------------------------------------------------------------- device storage
# Slab: the owner of one `jh_bvec*` -- a whole block vector (nblocks blocks in ONE hipMalloc), a stand-alone array (one
# block), or a view / wrap borrowing another slab's memory (`keep` holds what it borrows from: destroying a parent
# invalidates its views, include/jetship.h).
mutable struct Slab
    handle::Ptr{Cvoid}
    nblocks::Int
    keep::Any
    function Slab(handle::Ptr{Cvoid}, nblocks::Integer, keep=nothing)
        s = new(handle, nblocks, keep)
        finalizer(s -> ccall((:jh_bvec_destroy, LIB), Cint, (Ptr{Cvoid},), s.handle), s)
        s
    end
end

# zero-filled (zeros(R)) or not (Array(R) = Array{T,N}(undef, ...), src/Jets.jl:105; the zero fill of 64 GiB is 11 ms) -- ccall wants literal names
_bvec_create(lens::Vector{Int64}, ::Type{T}, h, undef::Bool) where {T} = undef ?
    ccall((:jh_bvec_create_uninit, LIB), Cint, (Int64, Ptr{Int64}, Cint, Ref{Ptr{Cvoid}}), length(lens), lens, dtype_code(T), h) :
    ccall((:jh_bvec_create, LIB), Cint, (Int64, Ptr{Int64}, Cint, Ref{Ptr{Cvoid}}), length(lens), lens, dtype_code(T), h)

# What a vector of 4 GiB or more is for (the library's knob alloc_role, include/jetship.h): 1 = an operator's output, 2 = data written once and
# read from then on, 0 = unknown.  Of the cached slabs of its size the library hands an output the one it has measured as fastest to write (the
# tall forward follows the slab it writes: 20.6-21.5 against 23.8-24.8 ms at the headline size), data the slowest (those read fastest).
const _alloc_role = Ref(0)
function with_alloc_role(f::Function, role::Integer)
    old = _alloc_role[]
    _alloc_role[] = role
    try
        return f()
    finally
        _alloc_role[] = old
    end
end

function _create(lens::Vector{Int64}, ::Type{T}, undef::Bool=false) where {T}
    _ensure_init()
    h = Ref{Ptr{Cvoid}}()
    hint = _alloc_role[] != 0 && sum(lens) * sizeof(T) >= (4 << 30)
    hint && tune!("alloc_role", _alloc_role[])
    st = _bvec_create(lens, T, h, undef)
    if st == 3                                                     # JH_ERR_NOMEM: Julia's GC does not see device memory -- unreachable vectors may
        GC.gc()                                                    # still hold theirs.  Their finalizers hand it to the library's slab cache; ask again.
        st = _bvec_create(lens, T, h, undef)
    end
    hint && tune!("alloc_role", 0)
    check(st)
    Slab(h[], length(lens))
end

function _view(parent::Slab, iblock::Integer)                      # block `iblock` (1-based) of a slab, by reference
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:jh_bvec_view, LIB), Cint, (Ptr{Cvoid}, Int64, Int64, Ref{Ptr{Cvoid}}), parent.handle, iblock - 1, 1, h))
    Slab(h[], 1, parent)
end

function _device_ptr(s::Slab)
    p = Ref{Ptr{Cvoid}}()
    check(ccall((:jh_bvec_info, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Cint}, Ref{Ptr{Cvoid}}), s.handle, C_NULL, C_NULL, C_NULL, p))
    p[]
end

# A dense column-major N-d array in HBM: a one-block jh_bvec.  `owner`/`iblock` say which block of which multi-block
# slab it is a view of (nothing / 0 for a stand-alone array): a BlockArray whose arrays are blocks 1..n of one owner is
# addressed as a whole through the owner's handle.
struct HipArray{T,N} <: AbstractArray{T,N}
    slab::Slab
    dims::NTuple{N,Int}
    owner::Union{Nothing,Slab}
    iblock::Int
end
HipArray{T,N}(slab::Slab, dims::NTuple{N,Int}) where {T,N} = HipArray{T,N}(slab, dims, nothing, 0)
handle(x::HipArray) = x.slab.handle

const DevBlockArray{T} = BlockArray{T,<:HipArray{T}}
const DevVec{T} = Union{HipArray{T},BlockArray{T,<:HipArray{T}}}

# the slab a device BlockArray lives in, or nothing when it was assembled from unrelated arrays (then the reference's
# per-block methods run, each block being a device array of its own)
function whole(x::BlockArray{T,<:HipArray{T}}) where {T}
    isempty(x.arrays) && return nothing
    o = x.arrays[1].owner
    (o === nothing || o.nblocks != length(x.arrays)) && return nothing
    for i in eachindex(x.arrays)
        (x.arrays[i].owner === o && x.arrays[i].iblock == i) || return nothing
    end
    o
end
handle(x::BlockArray{T,<:HipArray{T}}) where {T} = (o = whole(x); o === nothing ? C_NULL : o.handle)

# ---------------------------------------------------------------- device spaces
# JetSpace (src/Jets.jl:40-68) whose factories (105-108) allocate in HBM.
struct HipSpace{T,N} <: JetAbstractSpace{T,N}
    n::NTuple{N,Int}
end
HipSpace(::Type{T}, n::Vararg{Int,N}) where {T<:HipEltype,N} = HipSpace{T,N}(n)
HipSpace(::Type{T}, n::NTuple{N,Int}) where {T<:HipEltype,N} = HipSpace{T,N}(n)
Base.size(R::HipSpace) = R.n
Base.eltype(::Type{HipSpace{T,N}}) where {T,N} = T
Base.eltype(::Type{HipSpace{T}}) where {T} = T
Base.vec(R::HipSpace) = HipSpace(eltype(R), length(R))
Base.similar(R::HipSpace{T,N}, dims::NTuple{N,Int}) where {T,N} = HipSpace(T, dims)
Base.similar(R::HipSpace, dims::Int...) = similar(R, dims)
Jets.space(x::HipArray{T,N}) where {T,N} = HipSpace{T,N}(size(x))                      # src/Jets.jl:126

const _draws = Ref{UInt64}(0)                                                           # one counter-RNG stream per rand()/randn() call
_next_stream() = (_draws[] += 1; _draws[])
_fill_uniform!(s::Slab, seed, stream) = check(ccall((:jh_fill_uniform, LIB), Cint, (Ptr{Cvoid}, UInt64, UInt64, Int64), s.handle, seed, stream, 0))
_fill_normal!(s::Slab, seed, stream) = check(ccall((:jh_fill_normal, LIB), Cint, (Ptr{Cvoid}, UInt64, UInt64, Int64), s.handle, seed, stream, 0))
_fill!(s::Slab, a) = check(ccall((:jh_fill, LIB), Cint, (Ptr{Cvoid}, Cdouble, Cdouble), s.handle, real(a), imag(a)))

Base.zeros(R::HipSpace{T,N}) where {T,N} = HipArray{T,N}(_create(Int64[length(R)], T), size(R))
Base.Array(R::HipSpace{T,N}) where {T,N} = HipArray{T,N}(_create(Int64[length(R)], T, true), size(R))   # undef, like the reference's Array(R) (src/Jets.jl:105)
Base.ones(R::HipSpace{T}) where {T} = (x = Array(R); _fill!(x.slab, one(T)); x)                         # (every element is written: no zero fill first)
Base.rand(R::HipSpace; seed=1) = (x = Array(R); _fill_uniform!(x.slab, seed, _next_stream()); x)
Base.randn(R::HipSpace; seed=1) = (x = Array(R); _fill_normal!(x.slab, seed, _next_stream()); x)

# Array/zeros/ones/rand/randn(R::JetBSpace) (src/Jets.jl:922-924) for block spaces of device spaces: ONE slab, block i at
# element offset R.indices[i][1]-1, handed out as the reference's BlockArray of block views
function _blockarray(R::JetBSpace{T,<:HipSpace}, undef::Bool=false) where {T}
    o = _create(Int64[length(R.indices[i]) for i = 1:length(R.indices)], T, undef)
    arrays = [HipArray{T,ndims(R.spaces[i])}(_view(o, i), size(R.spaces[i]), o, i) for i = 1:length(R.spaces)]
    BlockArray(arrays, R.indices), o
end
Base.zeros(R::JetBSpace{T,S}) where {T,S<:HipSpace} = _blockarray(R)[1]
Base.Array(R::JetBSpace{T,S}) where {T,S<:HipSpace} = _blockarray(R, true)[1]
Base.ones(R::JetBSpace{T,S}) where {T,S<:HipSpace} = ((x, o) = _blockarray(R, true); _fill!(o, one(T)); x)
Base.rand(R::JetBSpace{T,S}; seed=1) where {T,S<:HipSpace} = ((x, o) = with_alloc_role(() -> _blockarray(R, true), 2); _fill_uniform!(o, seed, _next_stream()); x)
Base.randn(R::JetBSpace{T,S}; seed=1) where {T,S<:HipSpace} = ((x, o) = with_alloc_role(() -> _blockarray(R, true), 2); _fill_normal!(o, seed, _next_stream()); x)
# A*m = mul!(zeros(range(A)), A, m) (src/Jets.jl:399) with the output allocated as an OUTPUT (more specific than the reference's method in m only)
Base.:*(A::Jop, m::HipArray) = mul!(with_alloc_role(() -> zeros(range(A)), 1), A, m)
Base.:*(A::Jop, m::BlockArray{T,<:HipArray{T}}) where {T} = mul!(with_alloc_role(() -> zeros(range(A)), 1), A, m)

# ---------------------------------------------------------------- HipArray: array interface
Base.size(x::HipArray) = x.dims
Base.IndexStyle(::Type{<:HipArray}) = IndexLinear()
# scalar indexing works (printing, generic fallbacks) but moves one element over PCIe per call: the slow path, like
# BlockArray's own findfirst-based getindex (src/Jets.jl:820-827)
function Base.getindex(x::HipArray{T}, i::Int) where {T}
    out = Vector{T}(undef, 1)
    check(ccall((:jh_download, LIB), Cint, (Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}), handle(x), i - 1, 1, out))
    out[1]
end
function Base.setindex!(x::HipArray{T}, v, i::Int) where {T}
    check(ccall((:jh_upload, LIB), Cint, (Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}), handle(x), i - 1, 1, T[convert(T, v)]))
    v
end
Base.similar(x::HipArray{T,N}) where {T,N} = zeros(HipSpace{T,N}(size(x)))
Base.similar(x::HipArray, ::Type{S}, dims::Dims{N}) where {S<:HipEltype,N} = zeros(HipSpace{S,N}(dims))
Base.similar(x::HipArray, ::Type{S}, dims::Dims{N}) where {S,N} = Array{S,N}(undef, dims)        # Bool masks etc. stay on the host
# similar(x::BlockArray[, T]) (src/Jets.jl:829-832) must not fall apart into n separate allocations: keep ONE slab
Base.similar(x::BlockArray{S,<:HipArray{S}}, ::Type{T}) where {S,T<:HipEltype} = zeros(JetBSpace([HipSpace{T,ndims(a)}(size(a)) for a in x.arrays]))

# reshape shares memory (src/Jets.jl:38): a second one-block view of the same elements
function Base.reshape(x::HipArray{T}, dims::Dims{N}) where {T,N}
    prod(dims) == length(x) || throw(DimensionMismatch("new dimensions $(dims) must be consistent with array size $(length(x))"))
    HipArray{T,N}(_view(x.slab, 1), dims, nothing, 0)
end
# reshape(x, R::JetBSpace) (src/Jets.jl:1112): a flat device vector seen as a block vector, sharing memory
function Base.reshape(x::HipArray{T,1}, R::JetBSpace{T,<:HipSpace}) where {T}
    length(x) == length(R) || error("dimension mismatch, unable to reshape block array")
    lens = Int64[length(R.indices[i]) for i = 1:length(R.indices)]
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:jh_bvec_wrap, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Int64}, Cint, Ref{Ptr{Cvoid}}), _device_ptr(x.slab), length(lens), lens, dtype_code(T), h))
    o = Slab(h[], length(lens), x.slab)
    BlockArray([HipArray{T,ndims(R.spaces[i])}(_view(o, i), size(R.spaces[i]), o, i) for i = 1:length(R.spaces)], R.indices)
end

# host <-> device copies: convert(Array, x) (src/Jets.jl:862-868), copyto! both ways
function Base.copyto!(dst::Array{T}, src::HipArray{T}) where {T}
    length(dst) == length(src) || throw(DimensionMismatch("copyto!: $(length(dst)) vs $(length(src)) elements"))
    check(ccall((:jh_download, LIB), Cint, (Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}), handle(src), 0, length(src), dst))
    dst
end
function Base.copyto!(dst::HipArray{T}, src::Array{T}) where {T}
    length(dst) == length(src) || throw(DimensionMismatch("copyto!: $(length(dst)) vs $(length(src)) elements"))
    check(ccall((:jh_upload, LIB), Cint, (Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}), handle(dst), 0, length(src), src))
    dst
end
function Base.copyto!(dst::HipArray{T}, src::HipArray{T}) where {T}
    check(ccall((:jh_copy, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), handle(dst), handle(src)))
    dst
end
Base.copy(x::HipArray) = copyto!(similar(x), x)
Base.Array(x::HipArray{T,N}) where {T,N} = copyto!(Array{T,N}(undef, size(x)), x)
HipArray(a::Array{T,N}) where {T<:HipEltype,N} = copyto!(zeros(HipSpace{T,N}(size(a))), a)
function Base.convert(::Type{Array}, x::BlockArray{T,<:HipArray{T}}) where {T}
    h = handle(x)
    h == C_NULL && return invoke(convert, Tuple{Type{Array},BlockArray{T}}, Array, x)
    out = Vector{T}(undef, length(x))
    check(ccall((:jh_download, LIB), Cint, (Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}), h, 0, length(x), out))
    out
end

# a page-locked host array (jh_host_alloc) for those copies: no first-touch page faults under the DMA
function pinned_array(::Type{T}, dims::Integer...) where {T}
    p = Ref{Ptr{Cvoid}}()
    check(ccall((:jh_host_alloc, LIB), Cint, (Csize_t, Ref{Ptr{Cvoid}}), prod(dims) * sizeof(T), p))
    a = unsafe_wrap(Array, convert(Ptr{T}, p[]), dims; own=false)
    finalizer(_ -> ccall((:jh_host_free, LIB), Cint, (Ptr{Cvoid},), p[]), a)
    a
end

# getblock!(x, i, xblock) / setblock!(x, i, xblock) (src/Jets.jl:915-916) with HOST arrays: one DMA, no elementwise fallback;
# with device arrays: one device-to-device copy
function Jets.getblock!(x::BlockArray{T,<:HipArray{T}}, iblock, xblock::Array{T}) where {T}
    o = whole(x)
    o === nothing && return copyto!(xblock, x.arrays[iblock])
    length(xblock) == length(x.arrays[iblock]) || throw(DimensionMismatch("getblock!: block $(iblock) has $(length(x.arrays[iblock])) elements"))
    check(ccall((:jh_getblock_copy, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Cint), o.handle, iblock - 1, xblock, 0))
    xblock
end
function Jets.setblock!(x::BlockArray{T,<:HipArray{T}}, iblock, xblock::Array{T}) where {T}
    o = whole(x)
    o === nothing && return copyto!(x.arrays[iblock], xblock)
    length(xblock) == length(x.arrays[iblock]) || throw(DimensionMismatch("setblock!: block $(iblock) has $(length(x.arrays[iblock])) elements"))
    check(ccall((:jh_setblock_copy, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Cint), o.handle, iblock - 1, xblock, 0))
    x.arrays[iblock]
end
Jets.getblock!(x::BlockArray{T,<:HipArray{T}}, iblock, xblock::HipArray{T}) where {T} = copyto!(xblock, x.arrays[iblock])
Jets.setblock!(x::BlockArray{T,<:HipArray{T}}, iblock, xblock::HipArray{T}) where {T} = copyto!(x.arrays[iblock], xblock)
function Jets.setblock!(x::BlockArray{T,<:HipArray{T}}, iblock, a::Number) where {T}                             # test/runtests.jl:518-519
    o = whole(x)
    o === nothing && return fill!(x.arrays[iblock], a)
    check(ccall((:jh_setblock_fill, LIB), Cint, (Ptr{Cvoid}, Int64, Cdouble, Cdouble), o.handle, iblock - 1, real(a), imag(a)))
    x.arrays[iblock]
end

# ---------------------------------------------------------------- fill!, norm, dot, extrema (src/Jets.jl:834-885): one pass over the slab
Base.fill!(x::HipArray, a) = (_fill!(x.slab, a); x)
function Base.fill!(x::BlockArray{T,<:HipArray{T}}, a) where {T}
    o = whole(x)
    o === nothing ? invoke(fill!, Tuple{BlockArray,Any}, x, a) : (_fill!(o, a); x)
end
function _norm(h::Ptr{Cvoid}, p::Real, ::Type{T}) where {T}
    out = Ref{Cdouble}()
    check(ccall((:jh_norm, LIB), Cint, (Ptr{Cvoid}, Cdouble, Ref{Cdouble}), h, p, out))
    float(real(T))(out[])
end
LinearAlgebra.norm(x::HipArray{T}, p::Real=2) where {T} = _norm(handle(x), p, T)
function LinearAlgebra.norm(x::BlockArray{T,<:HipArray{T}}, p::Real=2) where {T}
    h = handle(x)
    h == C_NULL ? invoke(norm, Tuple{BlockArray{T},Real}, x, p) : _norm(h, p, T)
end
function _dot(hx::Ptr{Cvoid}, hy::Ptr{Cvoid}, ::Type{T}) where {T}
    re, im = Ref{Cdouble}(), Ref{Cdouble}()
    check(ccall((:jh_dot, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Cdouble}, Ref{Cdouble}), hx, hy, re, im))
    T <: Complex ? T(re[], im[]) : T(re[])
end
LinearAlgebra.dot(x::HipArray{T}, y::HipArray{T}) where {T} = _dot(handle(x), handle(y), T)
function LinearAlgebra.dot(x::BlockArray{T,<:HipArray{T}}, y::BlockArray{T,<:HipArray{T}}) where {T}
    hx, hy = handle(x), handle(y)
    (hx == C_NULL || hy == C_NULL) ? invoke(dot, Tuple{BlockArray{T},BlockArray{T}}, x, y) : _dot(hx, hy, T)
end

# the block-wise norms / inner products themselves, ALL blocks in one pass over the slab (round 6): what src/Jets.jl:836-846 / 850-856 compute block by
# block before they combine them -- `blocknorms(r)` instead of `[norm(getblock(r, i)) for i in 1:nblocks(r)]` (a launch and a host round trip per block)
function blocknorms(x::BlockArray{T,<:HipArray{T}}, p::Real=2) where {T}
    h = handle(x)
    h == C_NULL && return [norm(x.arrays[i], p) for i = 1:length(x.arrays)]
    out = Vector{Cdouble}(undef, length(x.arrays))
    check(ccall((:jh_norm_blocks, LIB), Cint, (Ptr{Cvoid}, Cdouble, Ptr{Cdouble}), h, p, out))
    float(real(T))[v for v in out]
end
function blockdots(x::BlockArray{T,<:HipArray{T}}, y::BlockArray{T,<:HipArray{T}}) where {T}
    hx, hy = handle(x), handle(y)
    (hx == C_NULL || hy == C_NULL) && return [dot(x.arrays[i], y.arrays[i]) for i = 1:length(x.arrays)]
    re, im = Vector{Cdouble}(undef, length(x.arrays)), Vector{Cdouble}(undef, length(x.arrays))
    check(ccall((:jh_dot_blocks, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}), hx, hy, re, im))
    T <: Complex ? T[T(re[i], im[i]) for i = 1:length(re)] : T[T(re[i]) for i = 1:length(re)]
end
function _extrema(h::Ptr{Cvoid}, ::Type{T}) where {T}
    mn, mx = Ref{Cdouble}(), Ref{Cdouble}()
    check(ccall((:jh_extrema, LIB), Cint, (Ptr{Cvoid}, Ref{Cdouble}, Ref{Cdouble}), h, mn, mx))
    T(mn[]), T(mx[])
end
Base.extrema(x::HipArray{T}) where {T<:Real} = _extrema(handle(x), T)
function Base.extrema(x::BlockArray{T,<:HipArray{T}}) where {T<:Real}
    h = handle(x)
    h == C_NULL ? invoke(extrema, Tuple{BlockArray{T}}, x) : _extrema(h, T)
end

# ---------------------------------------------------------------- broadcast: any elementwise expression, one fused pass
# copyto!(dest::BlockArray, bc::Broadcasted{BlockArrayStyle}) (src/Jets.jl:899-911) for device vectors: print the
# Broadcasted tree as a C expression over x0.. (vector leaves) and s0.. (scalar leaves), let libjetship compile it once
# with hiprtc (jh_bcast_compile) and stream the slabs in ONE kernel.  `a*u .+ b*v .+ c*w`, `exp.(-u.^2) .* v`, ...
struct HipStyle{N} <: Broadcast.AbstractArrayStyle{N} end
HipStyle(::Val{N}) where {N} = HipStyle{N}()
HipStyle{M}(::Val{N}) where {M,N} = HipStyle{N}()
Base.BroadcastStyle(::Type{<:HipArray{T,N}}) where {T,N} = HipStyle{N}()
Base.BroadcastStyle(a::BlockArrayStyle, ::HipStyle) = a                              # a flat device vector among block vectors (src/Jets.jl:902)
_find_hip(bc::Broadcast.Broadcasted) = _find_hip(bc.args)
_find_hip(args::Tuple) = _find_hip(_find_hip(args[1]), Base.tail(args))
_find_hip(x) = x
_find_hip(::Tuple{}) = nothing
_find_hip(a::HipArray, rest) = a
_find_hip(::Any, rest) = _find_hip(rest)
Base.similar(bc::Broadcast.Broadcasted{HipStyle{N}}, ::Type{T}) where {N,T} = similar(_find_hip(bc), T, map(length, axes(bc)))

const _cfun = Dict{Any,String}(+ => "+", - => "-", * => "*", / => "/", exp => "exp", log => "log", sqrt => "sqrt", sin => "sin",
                               cos => "cos", tanh => "tanh", abs => "abs", abs2 => "abs2", conj => "conj", real => "real",
                               imag => "imag", sign => "sign", max => "fmax", min => "fmin")
function _emit(x::DevVec, vecs, scals)
    i = findfirst(v -> v === x, vecs)
    i === nothing && (push!(vecs, x); i = length(vecs))
    "x$(i-1)"
end
_emit(x::Number, vecs, scals) = (push!(scals, x); "s$(length(scals)-1)")
_emit(x::Base.RefValue, vecs, scals) = _emit(x[], vecs, scals)
_emit(x, vecs, scals) = error("broadcast over device vectors: operand of type $(typeof(x)) is not supported (host arrays must be copied to the device first)")
function _emit(bc::Broadcast.Broadcasted, vecs, scals)
    bc.f === identity && return _emit(bc.args[1], vecs, scals)
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
const _bcast_programs = Dict{Tuple{String,DataType,Int,Int,Int},Ptr{Cvoid}}()
# bit k-1 set: vector leaf k is REAL of the matching precision in a complex broadcast (a real mask on a complex vector);
# bit length(vecs)+k-1: scalar k is a Real (a::Real * z works part by part: no 0 * Inf from an imaginary part it does not have)
_real_mask(::Type{T}, vecs, scals) where {T<:Real} = 0
_real_mask(::Type{Complex{R}}, vecs, scals) where {R} =
    sum(Int[1 << (k - 1) for k = 1:length(vecs) if eltype(vecs[k]) === R]) + sum(Int[1 << (length(vecs) + k - 1) for k = 1:length(scals) if scals[k] isa Real])
# bit k-1: scalar k is Float64-based against 32-bit elements (JH_SCALAR_WIDE): Julia promotes, computes in Float64 and rounds once on the store
_wide_mask(::Type{T}, scals) where {T} = real(T) === Float32 ? sum(Int[1 << (k - 1) for k = 1:length(scals) if real(typeof(scals[k])) === Float64]) : 0
function _bcast!(dest::DevVec{T}, bc::Broadcast.Broadcasted) where {T}
    vecs, scals = Any[], Number[]
    expr = _emit(bc, vecs, scals)
    hs = Ptr{Cvoid}[handle(v) for v in vecs]
    any(h -> h == C_NULL, hs) && error("broadcast over device block vectors that do not live in one slab")
    if expr == "x0" && isempty(scals)                              # dest .= x
        check(ccall((:jh_copy, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), handle(dest), hs[1]))
        return dest
    end
    mask = _real_mask(T, vecs, scals)
    wide = _wide_mask(T, scals)
    prog = get!(_bcast_programs, (expr, T, length(vecs), length(scals), mask | (wide << 32))) do
        h = Ref{Ptr{Cvoid}}()
        if wide != 0
            check(ccall((:jh_bcast_compile_typed, LIB), Cint, (Cstring, Cint, Cint, Cint, Cint, Cint, Ref{Ptr{Cvoid}}), expr, dtype_code(T), length(vecs), mask, length(scals), wide, h))
        elseif mask == 0
            check(ccall((:jh_bcast_compile, LIB), Cint, (Cstring, Cint, Cint, Cint, Ref{Ptr{Cvoid}}), expr, dtype_code(T), length(vecs), length(scals), h))
        else
            check(ccall((:jh_bcast_compile_mixed, LIB), Cint, (Cstring, Cint, Cint, Cint, Cint, Ref{Ptr{Cvoid}}), expr, dtype_code(T), length(vecs), mask, length(scals), h))
        end
        h[]
    end
    sc = Cdouble[]
    foreach(a -> (push!(sc, real(a)); push!(sc, imag(a))), scals)
    check(ccall((:jh_bcast_apply, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Ptr{Cdouble}), prog, handle(dest), hs, sc))
    dest
end
Base.copyto!(dest::HipArray{T}, bc::Broadcast.Broadcasted{<:HipStyle}) where {T} = _bcast!(dest, bc)
function Base.copyto!(dest::BlockArray{T,<:HipArray{T,N}}, bc::Broadcast.Broadcasted{BlockArrayStyle}) where {T,N}
    handle(dest) == C_NULL && return invoke(copyto!, Tuple{BlockArray{T,<:AbstractArray{T,N}},Broadcast.Broadcasted{BlockArrayStyle}}, dest, bc)
    _bcast!(dest, bc)
end

# ---------------------------------------------------------------- device-native operator kinds
# recognised by typeof(df!) exactly as iszero/isblockop do (src/Jets.jl:949, 1097)
_hadamard!(dst, x, y, flags) = (check(ccall((:jh_hadamard, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint), handle(dst), handle(x), handle(y), flags)); dst)
JopHipDiagonal_df!(d, m; diagonal, kwargs...) = _hadamard!(d, diagonal, m, 0)          # d .= diagonal .* m            (test/runtests.jl:3)
JopHipDiagonal_df′!(m, d; diagonal, kwargs...) = _hadamard!(m, diagonal, d, 1)         # m .= conj.(diagonal) .* d     (test/runtests.jl:4)
function JopHipDiagonal(diag::HipArray{T,N}) where {T,N}
    spc = HipSpace{T,N}(size(diag))
    JopLn(;df! = JopHipDiagonal_df!, df′! = JopHipDiagonal_df′!, dom = spc, rng = spc, s = (diagonal=diag,))
end
# weights over a block range (`W ∘ A`, `A' ∘ W ∘ A`): the diagonal is a device BlockArray in one slab, the operator lives on its block space
function JopHipDiagonal(diag::BlockArray{T,<:HipArray{T}}) where {T}
    spc = space(diag)
    JopLn(;df! = JopHipDiagonal_df!, df′! = JopHipDiagonal_df′!, dom = spc, rng = spc, s = (diagonal=diag,))
end

# the reference's nonlinear fixture JopBar (test/runtests.jl:19-24) on device vectors: kind SQUARE
JopHipSquare_f!(d, m; kwargs...) = _hadamard!(d, m, m, 0)                              # d .= m.^2
JopHipSquare_df!(δd, δm; mₒ, kwargs...) = _hadamard!(δd, mₒ, δm, 2)                    # δd .= 2 .* mₒ .* δm
JopHipSquare_df′!(δm, δd; mₒ, kwargs...) = _hadamard!(δm, mₒ, δd, 3)                   # δm .= conj.(2 .* mₒ) .* δd
JopHipSquare(spc::HipSpace) = JopNl(;f! = JopHipSquare_f!, df! = JopHipSquare_df!, df′! = JopHipSquare_df′!, dom = spc, rng = spc)

# the reference's dense fixture JopBaz (test/runtests.jl:27-33): d = A*m, m = A'*d for a column-major matrix in HBM
function _gemv!(y, A::HipArray{T,2}, x, adjoint) where {T}
    check(ccall((:jh_gemv, LIB), Cint, (Ptr{Cvoid}, Int64, Int64, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Cint),
                _device_ptr(A.slab), size(A, 1), size(A, 2), dtype_code(T), handle(y), handle(x), adjoint))
    y
end
JopHipDense_df!(d, m; A, kwargs...) = _gemv!(d, A, m, 0)
JopHipDense_df′!(m, d; A, kwargs...) = _gemv!(m, A, d, 1)
JopHipDense(A::HipArray{T,2}) where {T} = JopLn(;df! = JopHipDense_df!, df′! = JopHipDense_df′!, dom = HipSpace(T, size(A, 2)), rng = HipSpace(T, size(A, 1)), s = (A=A,))

struct jh_block_desc          # mirrors include/jetship.h
    kind::Int32
    adjoint::Int32
    coeff::Ptr{Cvoid}
    scale_re::Cdouble
    scale_im::Cdouble
    nr::Int64
    nc::Int64
    scale_flags::Int32        # JH_SCALAR_*: the TYPE of a, which Julia's `a * m` dispatches on
    reserved::Int32
end
jh_block_desc(kind, adjoint, coeff, re, im, nr, nc) = jh_block_desc(kind, adjoint, coeff, re, im, nr, nc, 0, 0)
# JH_SCALAR_COMPLEX = 1: a Complex scalar takes the full complex product even when its imaginary part is zero (a Real one multiplies part
# by part); JH_SCALAR_WIDE = 2: a Float64-based scalar against 32-bit elements is promoted arithmetic, rounded once on the store
_scalar_flags(a::Number) = Cint((a isa Complex ? 1 : 0) | (real(typeof(a)) === Float64 ? 2 : 0))

function block_desc(op::Jop)
    adj = op isa JopAdjoint
    base = adj ? op.op : op
    j = jet(base)
    nr, nc = length(range(base)), length(domain(base))
    if base isa JopNl
        return j.f! === JopHipSquare_f! ? jh_block_desc(5, 0, C_NULL, 0, 0, nr, nc) : nothing
    elseif j.df! === JopZeroBlock_df!
        return jh_block_desc(0, adj, C_NULL, 0, 0, nr, nc)
    elseif j.df! === JopHipDiagonal_df!
        return jh_block_desc(3, adj, _device_ptr(state(base).diagonal.slab), 0, 0, nr, nc)
    elseif j.df! === JopHipDense_df!
        return jh_block_desc(4, adj, _device_ptr(state(base).A.slab), 0, 0, nr, nc)
    elseif j.df! === JopHipSquare_df!                                   # JopLn(F) of a JopHipSquare: its Jacobian
        return jh_block_desc(5, adj, C_NULL, 0, 0, nr, nc)
    elseif j.df! === _constdiag_df!                                     # a*I, src/Jets.jl:1159-1164
        a = state(base).a
        return jh_block_desc(2, adj, C_NULL, real(a), imag(a), nr, nc, _scalar_flags(a), 0)
    end
    nothing            # not device-native: the reference's per-block loop handles it (on device arrays, child by child)
end

const _handles = IdDict{Any,Ptr{Cvoid}}()        # ops matrix -> jh_blockop* (C_NULL: has a child the device does not know)
const _points = IdDict{Any,Any}()                # ops matrix -> the mₒ the device operator borrows (kept alive here)

# several contexts in one session: make the context of the first coefficient array current before jh_blockop_create
function _enter_context_of_coefficients(ops)
    for op in ops
        base = op isa JopAdjoint ? op.op : op
        st = state(base)
        arr = hasproperty(st, :diagonal) ? st.diagonal : (hasproperty(st, :A) ? st.A : nothing)
        if arr isa HipArray
            c = Ref{Cint}(-1)
            check(ccall((:jh_bvec_context, LIB), Cint, (Ptr{Cvoid}, Ref{Cint}, Ptr{Cint}), arr.slab.handle, c, C_NULL))
            check(ccall((:jh_context_use, LIB), Cint, (Cint,), c[]))
            return
        end
    end
end

function native_handle(ops::AbstractMatrix{<:Jop}, ::Type{T}) where {T}
    get!(_handles, ops) do
        descs = [block_desc(ops[i,j]) for i = 1:size(ops,1), j = 1:size(ops,2)]      # column-major == the C layout
        any(isnothing, descs) && return C_NULL
        row_len = Int64[length(range(ops[i,1])) for i = 1:size(ops,1)]
        col_len = Int64[length(domain(ops[1,j])) for j = 1:size(ops,2)]
        h = Ref{Ptr{Cvoid}}()
        _enter_context_of_coefficients(ops)                                          # the operator lives where its coefficients live
        check(ccall((:jh_blockop_create, LIB), Cint, (Int64, Int64, Ptr{jh_block_desc}, Ptr{Int64}, Ptr{Int64}, Cint, Ref{Ptr{Cvoid}}),
                    size(ops,1), size(ops,2), vec(convert(Matrix{jh_block_desc}, descs)), row_len, col_len, dtype_code(T), h))
        h[]
    end
end

# the device handle of a tall (one block column) JopBlock, or C_NULL
function tall_native(A::Jop, ::Type{T}) where {T}
    (A isa JopLn && jet(A).df! === JetBlock_df! && size(state(A).ops, 2) == 1) || return C_NULL
    native_handle(state(A).ops, T)
end

# ---------------------------------------------------------------- the hot path: ONE ccall per mul!
# More specific methods of the reference's block loops (src/Jets.jl:988-1057) for device vectors; anything the device does
# not know (a user closure among the children, a BlockArray that is not one slab) falls through to the reference's loop.
function Jets.JetBlock_df!(d::BlockArray{T,<:HipArray{T}}, m::DevVec{T}; ops, dom, rng, kwargs...) where {T}
    h, hd, hm = native_handle(ops, T), handle(d), handle(m)
    (h == C_NULL || hd == C_NULL || hm == C_NULL) && return invoke(JetBlock_df!, Tuple{AbstractArray,AbstractArray}, d, m; ops=ops, dom=dom, rng=rng, kwargs...)
    check(ccall((:jh_blockop_mul, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), h, hd, hm))
    d
end

function Jets.JetBlock_df′!(m::DevVec{T}, d::BlockArray{T,<:HipArray{T}}; ops, dom, rng, kwargs...) where {T}
    h, hd, hm = native_handle(ops, T), handle(d), handle(m)
    (h == C_NULL || hd == C_NULL || hm == C_NULL) && return invoke(JetBlock_df′!, Tuple{AbstractArray,AbstractArray}, m, d; ops=ops, dom=dom, rng=rng, kwargs...)
    check(ccall((:jh_blockop_mul_adj, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), h, hm, hd))
    m
end

# nonlinear block operators: JetBlock_f! (src/Jets.jl:988-1008) and the block point! (1059-1066); the Jacobian then runs
# through the two methods above (the device operator borrows mₒ: it is kept alive in _points until the next point!)
function Jets.JetBlock_f!(d::BlockArray{T,<:HipArray{T}}, m::DevVec{T}; ops, dom, rng, kwargs...) where {T}
    h, hd, hm = native_handle(ops, T), handle(d), handle(m)
    (h == C_NULL || hd == C_NULL || hm == C_NULL) && return invoke(JetBlock_f!, Tuple{AbstractArray,AbstractArray}, d, m; ops=ops, dom=dom, rng=rng, kwargs...)
    check(ccall((:jh_blockop_f, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), h, hd, hm))
    d
end

function Jets.point!(j::Jet{D,R,typeof(JetBlock_f!)}, mₒ::DevVec{T}) where {D<:JetAbstractSpace,R<:JetAbstractSpace,T}
    invoke(point!, Tuple{Jet{D,R,typeof(JetBlock_f!)},AbstractArray}, j, mₒ)      # children first (1062-1064)
    ops = state(j).ops
    h, hm = native_handle(ops, T), handle(mₒ)
    if h != C_NULL && hm != C_NULL
        check(ccall((:jh_blockop_point, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), h, hm))
        _points[ops] = mₒ
    end
    j
end

# (A' ∘ A) * m fused (src/Jets.jl:530-534 over (A', A)): coefficients read once, no range-side temporary, the bits of the
# unfused chain.  Any other composite runs the reference's chain on device arrays.
# dot_product_test on device vectors (src/Jets.jl:1211-1226): the reference builds ones(domain(op)) / ones(range(op)) as default masks and
# the masked copies `mmask .* m`, `dmask .* d` -- at the headline size each of those is another 64 GiB next to A, d and A*m, more than the
# device holds.  `1 .* x` has x's bits, so without masks neither the ones nor the products are materialised; same values, same order.
function Jets.dot_product_test(op::JopLn, m::DevVec{T}, d::DevVec{T}; mmask=[], dmask=[]) where {T}
    mm = length(mmask) == 0 ? m : mmask .* m
    dd = length(dmask) == 0 ? d : dmask .* d
    ms = op' * dd
    lhs = dot(mm, ms)
    ds = op * mm
    rhs = dot(ds, dd)
    (eltype(lhs) <: Complex && eltype(rhs) <: Complex) ? (lhs, rhs) : (real(lhs), real(rhs))
end

function Jets.JetComposite_df!(d::HipArray{T}, m::HipArray{T}; ops, kwargs...) where {T}
    if length(ops) == 2 && ops[1] isa JopAdjoint && ops[1].op === ops[2]
        h = tall_native(ops[2], T)
        if h != C_NULL
            check(ccall((:jh_blockop_normal_mul, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), h, handle(d), handle(m)))
            return d
        end
    end
    _fused_chain!(d, m, _stages_df(ops), T, 0) && return d              # chains of any depth: every fusable run in one pass (round 6)
    invoke(JetComposite_df!, Tuple{AbstractArray,Any}, d, m; ops=ops, kwargs...)
end
# (A' ∘ A) on an N × K GRID of equal diagonals, K = 2 … 4 (a multi-parameter operator: domain and output are block vectors): one pass over the
# coefficients, the two-stage chain's bits (jh_grid_normal.hip); anything else: the reference's chain
function Jets.JetComposite_df!(d::BlockArray{T,<:HipArray{T}}, m::BlockArray{T,<:HipArray{T}}; ops, kwargs...) where {T}
    if length(ops) == 2 && ops[1] isa JopAdjoint && ops[1].op === ops[2] && ops[2] isa JopLn && jet(ops[2]).df! === JetBlock_df! && 2 <= size(state(ops[2]).ops, 2) <= 4
        h, hd, hm = native_handle(state(ops[2]).ops, T), handle(d), handle(m)
        if h != C_NULL && hd != C_NULL && hm != C_NULL
            st = ccall((:jh_blockop_normal_mul, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), h, hd, hm)
            st == 4 || (check(st); return d)                    # JH_ERR_UNSUPPORTED (blocks of several kinds, K > 4): the chain below
        end
    end
    invoke(JetComposite_df!, Tuple{AbstractArray,Any}, d, m; ops=ops, kwargs...)
end
# JetComposite_f! (524-528) on one device space: runs of elementwise stages (F ∘ A ∘ F ∘ A, benchmark/benchmarks.jl:73) in one JIT-compiled pass
function Jets.JetComposite_f!(d::HipArray{T}, m::HipArray{T}; ops, kwargs...) where {T}
    _fused_chain!(d, m, Any[ops[i] for i = length(ops):-1:1], T, 0) && return d
    invoke(JetComposite_f!, Tuple{AbstractArray,Any}, d, m; ops=ops, kwargs...)
end
# the adjoint of a domain -> domain composite (M' ∘ A' ∘ W ∘ A ∘ M is its own shape adjointed)
function Jets.JetComposite_df′!(m::HipArray{T}, d::HipArray{T}; ops, kwargs...) where {T}
    _fused_chain!(m, d, _stages_df′(ops), T, 0) && return m
    invoke(JetComposite_df′!, Tuple{AbstractArray,Any}, m, d; ops=ops, kwargs...)
end

# ---- scalar * operator and sums of tall device operators: ONE ccall per mul! (rows a18 / a19 of the scope table) --------------------------
# `a * A` (src/Jets.jl:1161-1164) builds its scalar stage on domain(A) for BOTH of its spaces, which composes only when A is square (the
# one case the reference tests, test/runtests.jl:789-795).  For a tall device operator the stage lives on range(A) instead -- the
# documented deviation of DESIGN.md section 5 -- so that `1.0*A1 - 2.0*A2 + 3.0*A3` (the reference's own example, 686) exists for the
# seismic-shot layout at all.
function Base.:*(a::Number, A::JopLn{<:Jet{<:HipSpace,<:JetBSpace}})
    _a = JopLn(dom = range(A), rng = range(A), df! = _constdiag_df!, df′! = _constdiag_df′!, s=(a=a,))
    _a ∘ A
end

# Scalars whose product with an element the fused kernels reproduce: Julia promotes (Float64 scalar, Float32 element) to Float64 and
# rounds once on the store -- JH_SCALAR_WIDE, the WIDE instantiations --, every other type here is converted to the element type first,
# T(a).  A Complex scalar (full complex product) and the big number types (BigFloat / BigInt promote the ELEMENT) take the unfused chain.
const FusableReal = Union{Float16,Float32,Float64,Base.BitInteger,Bool,AbstractIrrational}
function _fusable_scalar(op::Jop)          # the `a` of an un-adjointed `a*I` stage (1159-1162), or nothing
    (op isa JopLn && jet(op).df! === _constdiag_df!) || return nothing
    a = state(op).a
    a isa FusableReal ? a : nothing
end

# (a, A): d = a * (A m) in one pass, the bits of the chain (tmp = A m; d .= a * tmp)
function Jets.JetComposite_df!(d::BlockArray{T,<:HipArray{T}}, m::HipArray{T}; ops, kwargs...) where {T}
    if length(ops) == 2 && handle(d) != C_NULL
        a = _fusable_scalar(ops[1])
        h = a === nothing ? C_NULL : tall_native(ops[2], T)
        if h != C_NULL
            st = ccall((:jh_blockop_mul_scaled, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cint), h, handle(d), handle(m), Float64(a), _scalar_flags(a))
            st == 4 || (check(st); return d)                    # JH_ERR_UNSUPPORTED (rows of mixed kinds, ragged blocks): the chain below
        end
    end
    handle(d) != C_NULL && _fused_chain!(d, m, _stages_df(ops), T, 0) && return d      # W ∘ A ∘ M and deeper (round 6)
    invoke(JetComposite_df!, Tuple{AbstractArray,Any}, d, m; ops=ops, kwargs...)
end
# (a, A)' = A' o a': m = A' (conj(a) d) in one pass (conj(a) == a for a Real a)
function Jets.JetComposite_df′!(m::HipArray{T}, d::BlockArray{T,<:HipArray{T}}; ops, kwargs...) where {T}
    if length(ops) == 2 && handle(d) != C_NULL
        a = _fusable_scalar(ops[1])
        h = a === nothing ? C_NULL : tall_native(ops[2], T)
        if h != C_NULL
            st = ccall((:jh_blockop_mul_adj_scaled, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cint), h, handle(m), handle(d), Float64(a), _scalar_flags(a))
            st == 4 || (check(st); return m)
        end
    end
    handle(d) != C_NULL && _fused_chain!(m, d, _stages_df′(ops), T, 0) && return m     # M' ∘ A' ∘ W' and deeper (round 6)
    invoke(JetComposite_df′!, Tuple{AbstractArray,Any}, m, d; ops=ops, kwargs...)
end

# JetSum_df! / JetSum_df′! (src/Jets.jl:639-655; the terms and signs arrive flattened, 657-676): when every term is a tall device-native
# operator A_k or the composite (a_k, A_k) of a fusable Real scalar, the whole sum is ONE ccall -- (K + 1) range-sized streams where the
# reference's loop (a zeros() temporary, one mul! and one accumulate pass per term) moves 5K + 1 -- with the chain's rounding sequence:
# product, scalar stage (typed: JH_SCALAR_WIDE for a Float64 scalar on Float32 elements), signed add, terms in order.
function _sum_term(op::Jop, ::Type{T}) where {T}          # (handle, scale, JH_SCALAR_* flags) or nothing
    op isa JopAdjoint && return nothing
    L = JopLn(op)
    if jet(L).df! === JetComposite_df!
        inner = state(L).ops
        length(inner) == 2 || return nothing
        a = _fusable_scalar(inner[1])
        a === nothing && return nothing
        h = tall_native(inner[2], T)
        return h == C_NULL ? nothing : (h, Float64(a), _scalar_flags(a))
    end
    h = tall_native(L, T)
    h == C_NULL ? nothing : (h, 1.0, Cint(0))
end
function _fused_sum(out, x, ops, sgns, ::Type{T}, transposed::Bool) where {T}      # true when the fused launch ran
    (handle(out) == C_NULL || handle(x) == C_NULL || length(ops) > 4096) && return false
    terms = map(op -> _sum_term(op, T), collect(ops))
    any(isnothing, terms) && return false
    hs = Ptr{Cvoid}[t[1] for t in terms]
    sc = Cdouble[t[2] for t in terms]
    fl = Int32[t[3] for t in terms]
    sg = Cdouble[s === (-) ? -1.0 : 1.0 for s in sgns]
    st = transposed ?
        ccall((:jh_blocksum_mul_adj_typed, LIB), Cint, (Cint, Ptr{Ptr{Cvoid}}, Ptr{Cdouble}, Ptr{Int32}, Ptr{Cdouble}, Ptr{Cvoid}, Ptr{Cvoid}), length(hs), hs, sc, fl, sg, handle(out), handle(x)) :
        ccall((:jh_blocksum_mul_typed, LIB), Cint, (Cint, Ptr{Ptr{Cvoid}}, Ptr{Cdouble}, Ptr{Int32}, Ptr{Cdouble}, Ptr{Cvoid}, Ptr{Cvoid}), length(hs), hs, sc, fl, sg, handle(out), handle(x))
    st == 4 && return false                                      # JH_ERR_UNSUPPORTED: the reference's loop
    check(st)
    true
end
function Jets.JetSum_df!(d::BlockArray{T,<:HipArray{T}}, m::HipArray{T}; ops, sgns, kwargs...) where {T}
    _fused_sum(d, m, ops, sgns, T, false) && return d
    _chain_sum!(d, m, ops, sgns, T, false) && return d                  # terms that are chains (W1 ∘ A1 - W2 ∘ A2 + ...): round 6
    invoke(JetSum_df!, Tuple{Any,Any}, d, m; ops=ops, sgns=sgns, kwargs...)
end
function Jets.JetSum_df′!(m::HipArray{T}, d::BlockArray{T,<:HipArray{T}}; ops, sgns, kwargs...) where {T}
    _fused_sum(m, d, ops, sgns, T, true) && return m
    _chain_sum!(m, d, ops, sgns, T, true) && return m
    invoke(JetSum_df′!, Tuple{Any,Any}, m, d; ops=ops, sgns=sgns, kwargs...)
end
# sums on the domain (A' ∘ W ∘ A + λ²I - B' ∘ B, src/Jets.jl:639-655 over composite terms): domain -> domain both ways
function Jets.JetSum_df!(d::HipArray{T}, m::HipArray{T}; ops, sgns, kwargs...) where {T}
    _chain_sum!(d, m, ops, sgns, T, false) && return d
    invoke(JetSum_df!, Tuple{Any,Any}, d, m; ops=ops, sgns=sgns, kwargs...)
end
function Jets.JetSum_df′!(m::HipArray{T}, d::HipArray{T}; ops, sgns, kwargs...) where {T}
    _chain_sum!(m, d, ops, sgns, T, true) && return m
    invoke(JetSum_df′!, Tuple{Any,Any}, m, d; ops=ops, sgns=sgns, kwargs...)
end

# ---------------------------------------------------------------- fused chains of ANY depth (round 6; the twin of jets.jl_amd/chains.py)
# The reference applies a composite stage by stage, right to left, each stage into a fresh zeros(range(op_i)) (src/Jets.jl:524-540).  Around a tall
# device operator A every other device-native stage is elementwise -- `a *` (1159-1164), a diagonal on the domain, a diagonal on the range -- so a
# maximal run of such stages is ONE call of the chain kernels (include/jetship.h jh_chain_*):
#     W ∘ A ∘ M            JH_CHAIN_FORWARD   d_i = R(a_i .* P(m))
#     M' ∘ A' ∘ W'         JH_CHAIN_ADJOINT   m   = Q(Σ_i conj(a_i) .* R(d_i))
#     M' ∘ A' ∘ W ∘ A ∘ M  JH_CHAIN_NORMAL    y   = Q(Σ_i conj(a_i) .* R(a_i .* P(m)))       (weighted / preconditioned normal equations)
# with the bits of the stage-by-stage chain (each stage's product rounded before the next reads it).  A stage the device does not know splits the
# chain: the runs on either side are fused, the stage runs on its own.
struct jh_chain_stage         # mirrors include/jetship.h
    kind::Int32               # 1 JH_STAGE_SCALE, 2 JH_STAGE_DIAG
    flags::Int32              # SCALE: JH_SCALAR_* / DIAG: 4 = JH_STAGE_CONJ
    a::Cdouble
    coeff::Ptr{Ptr{Cvoid}}    # DIAG: device pointers, one per block row on the range side, one on the domain side
    row_flags::Ptr{UInt8}
end
const _CHAIN_MAX_STAGES = 4
_stages_df(ops) = Any[JopLn(ops[i]) for i = length(ops):-1:1]            # application order (530-534)
_stages_df′(ops) = Any[JopLn(ops[i])' for i = 1:length(ops)]             # (536-540)

# what a stage is: (kind = :tall / :scale / :diag / :identity / :opaque, ...)
function _chain_stage(op::Jop, ::Type{T}) where {T}
    adj = op isa JopAdjoint
    base = adj ? op.op : op
    if base isa JopNl                                                   # a nonlinear stage of JetComposite_f! (524-528): its f!
        return (!adj && jet(base).f! === JopHipSquare_f!) ? (kind=:square_f, op=op) : (kind=:opaque, op=op)
    end
    base isa JopLn || return (kind=:opaque, op=op)
    j = jet(base)
    if j.df! === JopHipSquare_df! && length(j.mₒ) == length(domain(base))   # the Jacobian of d .= m.^2 about mₒ: δd .= 2 .* mₒ .* δm
        return (kind=:square_df, op=op, vec=j.mₒ, conj=adj)
    end
    if j.df! === _constdiag_df!
        a = state(base).a
        a isa FusableReal || return (kind=:opaque, op=op)               # a Complex scalar takes the full product: the stage-by-stage chain
        return (kind=:scale, op=op, a=Float64(a), flags=(sizeof(real(T)) == 8 ? Cint(0) : _scalar_flags(a)))
    elseif j.df! === JopHipDiagonal_df!
        return (kind=:diag, op=op, vec=state(base).diagonal, conj=adj)
    end
    h = tall_native(base, T)
    (h != C_NULL && size(state(base).ops, 1) >= 2) && return (kind=:tall, op=op, h=h, adj=adj, nrow=size(state(base).ops, 1), n=length(domain(base)))
    (kind=:opaque, op=op)
end
_elementwise(st) = st.kind === :scale || st.kind === :diag || st.kind === :identity
_active(sts) = [st for st in sts if st.kind !== :identity]

# cut the stages into steps: (:chain, type, tall, pre, mid, post, first, last) or (:op, index)   (chains.py: _segments)
function _chain_segments(st::Vector)
    steps, i, n = Any[], 1, length(st)
    while i <= n
        j = i
        while j <= n && _elementwise(st[j]); j += 1; end
        if j <= n && st[j].kind === :tall
            t = st[j]
            if !t.adj                                                   # E* A E* [A' E*]
                while length(_active(st[i:j-1])) > _CHAIN_MAX_STAGES
                    push!(steps, (:op, i)); i += 1
                end
                pre = _active(st[i:j-1])
                k = j + 1
                while k <= n && _elementwise(st[k]) && length(_active(st[j+1:k])) <= _CHAIN_MAX_STAGES; k += 1; end
                mid = _active(st[j+1:k-1])
                if k <= n && st[k].kind === :tall && st[k].adj && st[k].h == t.h
                    l = k + 1
                    while l <= n && _elementwise(st[l]) && length(_active(st[k+1:l])) <= _CHAIN_MAX_STAGES; l += 1; end
                    push!(steps, (:chain, Cint(2), t, pre, mid, _active(st[k+1:l-1]), i, l - 1)); i = l
                elseif !isempty(pre) || !isempty(mid)
                    push!(steps, (:chain, Cint(0), t, pre, mid, Any[], i, k - 1)); i = k
                else
                    push!(steps, (:op, j)); i = j + 1
                end
            else                                                        # E* A' E*
                while length(_active(st[i:j-1])) > _CHAIN_MAX_STAGES
                    push!(steps, (:op, i)); i += 1
                end
                mid = _active(st[i:j-1])
                l = j + 1
                while l <= n && _elementwise(st[l]) && length(_active(st[j+1:l])) <= _CHAIN_MAX_STAGES; l += 1; end
                post = _active(st[j+1:l-1])
                if !isempty(mid) || !isempty(post)
                    push!(steps, (:chain, Cint(1), t, Any[], mid, post, i, l - 1)); i = l
                else
                    push!(steps, (:op, j)); i = j + 1
                end
            end
        else
            for q = i:max(j - 1, i); push!(steps, (:op, q)); end
            i = max(j, i + 1)
        end
    end
    steps
end

# Runs of elementwise stages with NO tall operator to lean on -- the reference's own composition benchmark G = F ∘ A ∘ F ∘ A with F: d .= m.^2 and A a
# diagonal (benchmark/benchmarks.jl:33-38, 55-60, 73-80), its Jacobian and its adjoint -- become ONE lazy Broadcasted tree, which _bcast! prints as a C
# expression and runs as one JIT-compiled pass (src/Jets.jl:889-911): four passes through three temporaries in the reference.
_bcast_kind(st) = st.kind === :scale || st.kind === :diag || st.kind === :identity || st.kind === :square_f || st.kind === :square_df
function _merge_bcast(steps::Vector, st::Vector)
    out, run = Any[], Int[]
    function flush()
        if length([q for q in run if st[q].kind !== :identity]) >= 2
            push!(out, (:bcast, run[1], run[end]))
        else
            for q in run; push!(out, (:op, q)); end
        end
        empty!(run)
    end
    for step in steps
        if step[1] === :op && _bcast_kind(st[step[2]]) && (isempty(run) || length(range(st[step[2]].op)) == length(range(st[run[1]].op)))
            push!(run, step[2])
        else
            flush()
            (step[1] === :op && _bcast_kind(st[step[2]])) ? push!(run, step[2]) : push!(out, step)
        end
    end
    flush()
    out
end
function _bcast_tree(st::Vector, first::Int, stop::Int, x)
    e = x
    for q = first:stop
        s = st[q]
        if s.kind === :scale
            base = s.op isa JopAdjoint ? s.op.op : s.op
            e = Broadcast.broadcasted(*, state(base).a, e)                 # the scalar as given: its TYPE decides the arithmetic (_wide_mask / _real_mask)
        elseif s.kind === :diag
            e = Broadcast.broadcasted(*, s.conj ? Broadcast.broadcasted(conj, s.vec) : s.vec, e)
        elseif s.kind === :square_f
            e = Broadcast.broadcasted(*, e, e)
        elseif s.kind === :square_df
            c2 = Broadcast.broadcasted(+, s.vec, s.vec)
            e = Broadcast.broadcasted(*, s.conj ? Broadcast.broadcasted(conj, c2) : c2, e)
        end
    end
    e
end

# a diagonal before A / after A' lives on the domain (n elements), one after A / before A' on the range (nrow * n)
function _chain_sides_ok(t, pre, mid, post)
    for st in vcat(pre, post); (st.kind === :diag && length(st.vec) != t.n) && return false; end
    for st in mid; (st.kind === :diag && length(st.vec) != t.nrow * t.n) && return false; end
    true
end

const _chain_handles = Dict{Any,Any}()          # signature -> (jh_chain* or C_NULL when the library declined, what the handle borrows)
_stage_sig(st) = st.kind === :scale ? (:s, st.a, st.flags) : (:d, UInt(_device_ptr(st.vec isa HipArray ? st.vec.slab : whole(st.vec))), length(st.vec), st.conj)

function _chain_handle(ctype::Cint, t, pre, mid, post, ::Type{T}) where {T}
    key = (ctype, t.h, map(_stage_sig, pre), map(_stage_sig, mid), map(_stage_sig, post))
    hit = get(_chain_handles, key, nothing)
    hit === nothing || return hit[1]
    keep = Any[]
    function pack(sts, nptr)
        arr = jh_chain_stage[]
        for st in sts
            if st.kind === :scale
                push!(arr, jh_chain_stage(1, st.flags, st.a, C_NULL, C_NULL))
            else
                base = _device_ptr(st.vec isa HipArray ? st.vec.slab : whole(st.vec))
                ptrs = Ptr{Cvoid}[base + (i - 1) * t.n * sizeof(T) for i = 1:nptr]
                push!(keep, ptrs); push!(keep, st.vec)
                push!(arr, jh_chain_stage(2, st.conj ? 4 : 0, 0.0, pointer(ptrs), C_NULL))
            end
        end
        push!(keep, arr)
        arr
    end
    a_pre, a_mid, a_post = pack(pre, 1), pack(mid, t.nrow), pack(post, 1)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    st = GC.@preserve keep ccall((:jh_chain_create, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{jh_chain_stage}, Cint, Ptr{jh_chain_stage}, Cint, Ptr{jh_chain_stage}, Ref{Ptr{Cvoid}}),
                                 t.h, ctype, length(a_pre), a_pre, length(a_mid), a_mid, length(a_post), a_post, h)
    st == 4 || check(st)                                                # JH_ERR_UNSUPPORTED: remembered, the run is applied stage by stage
    _chain_handles[key] = (st == 4 ? C_NULL : h[], keep)
    _chain_handles[key][1]
end
function close_chains!()                                                # release every chain handle (their operators are about to be closed)
    for (h, _) in values(_chain_handles)
        h == C_NULL || ccall((:jh_chain_destroy, LIB), Cint, (Ptr{Cvoid},), h)
    end
    empty!(_chain_handles)
end

# x -> stages -> out with every fusable run in one ccall; false when nothing fuses (the caller runs the reference's chain).
# accumulate != 0 (a term of a sum, 634/643/652): only when the WHOLE chain is one fused run.
function _fused_chain!(out, x, stages::Vector, ::Type{T}, accumulate::Integer) where {T}
    (handle(out) == C_NULL || handle(x) == C_NULL) && return false
    st = Any[_chain_stage(op, T) for op in stages]
    steps = Any[]
    for step in _chain_segments(st)
        if step[1] === :chain && !_chain_sides_ok(step[3], step[4], step[5], step[6])
            for q = step[7]:step[8]; push!(steps, (:op, q)); end
        else
            push!(steps, step)
        end
    end
    steps = _merge_bcast(steps, st)
    any(s -> s[1] === :chain || s[1] === :bcast, steps) || return false
    (accumulate != 0 && (length(steps) != 1 || steps[1][1] !== :chain)) && return false
    cur = x
    for (k, step) in enumerate(steps)
        last = k == length(steps)
        if step[1] === :op
            op = st[step[2]].op
            cur = mul!(last ? fill!(out, 0) : zeros(range(op)), op, cur)
            continue
        end
        if step[1] === :bcast
            dst = last ? out : zeros(range(st[step[3]].op))
            cur = _bcast!(dst, _bcast_tree(st, step[2], step[3], cur))
            continue
        end
        _, ctype, t, pre, mid, post, first, stop = step
        h = _chain_handle(ctype, t, pre, mid, post, T)
        dst = last ? out : zeros(range(st[stop].op))
        status = h == C_NULL ? Cint(4) : ccall((:jh_chain_apply, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint), h, handle(dst), handle(cur), last ? accumulate : 0)
        if status == 4                                                  # the library declined this run (many rows of small blocks, ...): its stages one by one
            accumulate != 0 && return false
            for q = first:stop
                op = st[q].op
                cur = mul!((last && q == stop) ? fill!(out, 0) : zeros(range(op)), op, cur)
            end
            continue
        end
        check(status)
        cur = dst
    end
    true
end

# JetSum with terms that are chains (src/Jets.jl:639-655): a term that is ONE fusable run adds itself to the output in its own last stage
# (jh_chain_apply(accumulate): +-2 the first term after `d .= 0`, +-1 afterwards); the other terms go through the temporary as in the reference.
# A bare block operator with a zero block is refused: the reference's shared temporary keeps the PREVIOUS term's row there (1022).
function _chain_term_fusable(op::Jop, ::Type{T}, transposed::Bool) where {T}
    L = JopLn(op)
    L isa JopAdjoint && return nothing
    inner = jet(L).df! === JetComposite_df! ? state(L).ops : (L,)
    stages = transposed ? _stages_df′(inner) : _stages_df(inner)
    steps = _chain_segments(Any[_chain_stage(o, T) for o in stages])
    (length(steps) == 1 && steps[1][1] === :chain) ? stages : nothing
end
function _chain_sum!(out, x, ops, sgns, ::Type{T}, transposed::Bool) where {T}
    (handle(out) == C_NULL || handle(x) == C_NULL) && return false
    plans = Any[_chain_term_fusable(op, T, transposed) for op in ops]
    any(p -> p !== nothing, plans) || return false
    for (op, p) in zip(ops, plans)                                       # an unfused term must overwrite the whole temporary: composites and sums do
        p === nothing && !(jet(JopLn(op)).df! === JetComposite_df! || jet(JopLn(op)).df! === JetSum_df!) && return false
    end
    started, tmp = false, nothing
    for (op, sg, p) in zip(ops, sgns, plans)
        sign = sg === (-) ? -1 : 1
        if p !== nothing && _fused_chain!(out, x, p, T, started ? sign : 2 * sign)
            started = true
            continue
        end
        started || fill!(out, 0)
        started = true
        tmp === nothing && (tmp = zeros(transposed ? domain(ops[1]) : range(ops[1])))
        broadcast!(sg, out, out, mul!(tmp, transposed ? JopLn(op)' : JopLn(op), x))
    end
    true
end

# ---------------------------------------------------------------- solver steps and the multi-GPU exchange
# u <- alpha*(A v) + beta*u, returns ||u||   /   v <- alpha*(A' (in_scale*u)) + beta*v, returns ||v||   (LSQR / CGLS halves)
function mul_axpby!(u::BlockArray{T,<:HipArray{T}}, A::JopLn, v::HipArray{T}, alpha::Real, beta::Real) where {T}
    nrm2 = Ref{Cdouble}()
    check(ccall((:jh_blockop_mul_axpby, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Ref{Cdouble}),
                tall_native(A, T), handle(u), handle(v), alpha, beta, nrm2))
    sqrt(nrm2[])
end
function mul_adj_axpby!(v::HipArray{T}, A::JopLn, u::BlockArray{T,<:HipArray{T}}, alpha::Real, beta::Real; in_scale::Real=1.0) where {T}
    nrm2 = Ref{Cdouble}()
    check(ccall((:jh_blockop_mul_adj_axpby, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Cdouble, Ref{Cdouble}),
                tall_native(A, T), handle(v), handle(u), alpha, beta, in_scale, nrm2))
    sqrt(nrm2[])
end

# d <- a * (A m)   /   m <- A' (conj(a) d) for a Real scalar of any type: the scalar-times-operator chain (src/Jets.jl:1159-1164) in one pass each
# way with the scalar's TYPE (a Float64 against Float32 elements: promoted product, one rounding -- the bits of `d .= a * tmp`).  The explicit
# spelling of what `mul!(d, a*A, m)` / `mul!(m, (a*A)', d)` reach through the JetComposite_df! / df′! methods above.
function mul_scaled!(d::BlockArray{T,<:HipArray{T}}, a::Real, A::JopLn, m::HipArray{T}) where {T}
    check(ccall((:jh_blockop_mul_scaled, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cint), tall_native(A, T), handle(d), handle(m), a, _scalar_flags(a)))
    d
end
function mul_adj_scaled!(m::HipArray{T}, a::Real, A::JopLn, d::BlockArray{T,<:HipArray{T}}) where {T}
    check(ccall((:jh_blockop_mul_adj_scaled, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cint), tall_native(A, T), handle(m), handle(d), a, _scalar_flags(a)))
    m
end

# one whole Golub-Kahan step in one pass: u <- alpha*(A v) + beta*u ; w <- A'u ; returns ||u||   (3/5 of the bytes of the two halves)
function bidiag_step!(u::BlockArray{T,<:HipArray{T}}, w::HipArray{T}, A::JopLn, v::HipArray{T}, alpha::Real, beta::Real) where {T}
    nrm2 = Ref{Cdouble}()
    check(ccall((:jh_blockop_bidiag_step, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Ref{Cdouble}),
                tall_native(A, T), handle(u), handle(v), handle(w), alpha, beta, nrm2))
    sqrt(nrm2[])
end

# the whole LSQR loop behind the ABI; `partitioned=true` is the row-partitioned solve (A, b = this rank's block rows, a
# collective over comm_init's communicator: every rank calls it) -- a communicator being alive never makes a local solve
# collective.  For operators the device does not know, IterativeSolvers.lsqr(vec(A), vec(b)) runs unchanged on device arrays
struct jh_lsqr_result; istop::Int32; itn::Int32; r1norm::Cdouble; r2norm::Cdouble; anorm::Cdouble; acond::Cdouble; arnorm::Cdouble; xnorm::Cdouble; end
function hip_lsqr!(x::HipArray{T}, A::JopLn, b::BlockArray{T,<:HipArray{T}}; x0::Bool=false, damp=0.0, atol=1e-6, btol=1e-6, conlim=1e8, maxiter=100, partitioned::Bool=false) where {T}
    h = tall_native(A, T)
    h == C_NULL && error("hip_lsqr!: needs a tall block operator of device-native children")
    res = Ref{jh_lsqr_result}()
    hist = Vector{Cdouble}(undef, 2 * maxiter)
    if partitioned
        check(ccall((:jh_lsqr_solve_partitioned, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint, Ref{jh_lsqr_result}, Ptr{Cdouble}),
                    h, handle(b), handle(x), x0, damp, atol, btol, conlim, maxiter, 0, res, hist))
    else
        check(ccall((:jh_lsqr_solve, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint, Ref{jh_lsqr_result}, Ptr{Cdouble}),
                    h, handle(b), handle(x), x0, damp, atol, btol, conlim, maxiter, 0, res, hist))   # b is overwritten (it becomes u)
    end
    x, res[], reshape(hist, 2, :)[:, 1:res[].itn]
end

# CGLS (conjugate gradients on the normal equations) behind the ABI: two passes per iteration, no range-sized temporary; b is
# overwritten (it becomes r = b - A x).  `partitioned=true`: the row-partitioned solve, a collective like hip_lsqr!'s
function hip_cgls!(x::HipArray{T}, A::JopLn, b::BlockArray{T,<:HipArray{T}}; x0::Bool=false, damp=0.0, atol=1e-6, btol=1e-6, maxiter=100, partitioned::Bool=false) where {T}
    h = tall_native(A, T)
    h == C_NULL && error("hip_cgls!: needs a tall block operator of device-native children")
    res = Ref{jh_lsqr_result}()
    hist = Vector{Cdouble}(undef, 2 * max(maxiter, 1))
    if partitioned
        check(ccall((:jh_cgls_solve_partitioned, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cdouble, Cdouble, Cdouble, Cint, Cint, Ref{jh_lsqr_result}, Ptr{Cdouble}),
                    h, handle(b), handle(x), x0, damp, atol, btol, maxiter, 0, res, hist))
    else
        check(ccall((:jh_cgls_solve, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cdouble, Cdouble, Cdouble, Cint, Cint, Ref{jh_lsqr_result}, Ptr{Cdouble}),
                    h, handle(b), handle(x), x0, damp, atol, btol, maxiter, 0, res, hist))
    end
    x, res[], reshape(hist, 2, :)[:, 1:res[].itn]
end
# CG on the normal equations through the fused A'A (one pass over the coefficients per iteration; b is only read)
function hip_cgnr!(x::HipArray{T}, A::JopLn, b::BlockArray{T,<:HipArray{T}}; x0::Bool=false, damp=0.0, atol=1e-6, btol=1e-6, maxiter=100, partitioned::Bool=false) where {T}
    h = tall_native(A, T)
    h == C_NULL && error("hip_cgnr!: needs a tall block operator of device-native children")
    res = Ref{jh_lsqr_result}()
    hist = Vector{Cdouble}(undef, 2 * max(maxiter, 1))
    if partitioned
        check(ccall((:jh_cgnr_solve_partitioned, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cdouble, Cdouble, Cdouble, Cint, Cint, Ref{jh_lsqr_result}, Ptr{Cdouble}),
                    h, handle(b), handle(x), x0, damp, atol, btol, maxiter, 0, res, hist))
    else
        check(ccall((:jh_cgnr_solve, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cdouble, Cdouble, Cdouble, Cint, Cint, Ref{jh_lsqr_result}, Ptr{Cdouble}),
                    h, handle(b), handle(x), x0, damp, atol, btol, maxiter, 0, res, hist))
    end
    x, res[], reshape(hist, 2, :)[:, 1:res[].itn]
end
function hip_cgnr_team!(xs::Vector{<:HipArray{T}}, As::Vector{<:JopLn}, bs::Vector{<:BlockArray{T,<:HipArray{T}}}; x0::Bool=false, damp=0.0, atol=1e-6,
                        btol=1e-6, maxiter=100) where {T}
    hs = Ptr{Cvoid}[tall_native(A, T) for A in As]
    any(h -> h == C_NULL, hs) && error("hip_cgnr_team!: every member needs a device-native tall block operator")
    res = Ref{jh_lsqr_result}()
    hist = Vector{Cdouble}(undef, 2 * max(maxiter, 1))
    check(ccall((:jh_cgnr_solve_team, LIB), Cint, (Cint, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Cint, Cdouble, Cdouble, Cdouble, Cint, Cint, Ref{jh_lsqr_result}, Ptr{Cdouble}),
                length(As), hs, Ptr{Cvoid}[handle(b) for b in bs], Ptr{Cvoid}[handle(x) for x in xs], x0 ? 1 : 0, damp, atol, btol, maxiter, 0, res, hist))
    xs, res[], reshape(hist, 2, :)[:, 1:res[].itn]
end
function hip_cgls_team!(xs::Vector{<:HipArray{T}}, As::Vector{<:JopLn}, bs::Vector{<:BlockArray{T,<:HipArray{T}}}; x0::Bool=false, damp=0.0, atol=1e-6,
                        btol=1e-6, maxiter=100) where {T}
    hs = Ptr{Cvoid}[tall_native(A, T) for A in As]
    any(h -> h == C_NULL, hs) && error("hip_cgls_team!: every member needs a device-native tall block operator")
    res = Ref{jh_lsqr_result}()
    hist = Vector{Cdouble}(undef, 2 * max(maxiter, 1))
    check(ccall((:jh_cgls_solve_team, LIB), Cint, (Cint, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Cint, Cdouble, Cdouble, Cdouble, Cint, Cint, Ref{jh_lsqr_result}, Ptr{Cdouble}),
                length(As), hs, Ptr{Cvoid}[handle(b) for b in bs], Ptr{Cvoid}[handle(x) for x in xs], x0 ? 1 : 0, damp, atol, btol, maxiter, 0, res, hist))
    xs, res[], reshape(hist, 2, :)[:, 1:res[].itn]
end

# one process per GPU: rank 0 makes the id, the host (MPI.jl, sockets, a file) ships it, every rank joins
comm_available() = ccall((:jh_comm_available, LIB), Cint, ()) == 0      # an RCCL can be loaded (no side effects)
comm_unique_id() = (id = Vector{UInt8}(undef, 128); check(ccall((:jh_comm_unique_id, LIB), Cint, (Ptr{UInt8},), id)); id)
comm_init(id::Vector{UInt8}, nranks::Integer, rank::Integer) = check(ccall((:jh_comm_init_rank, LIB), Cint, (Ptr{UInt8}, Cint, Cint), id, nranks, rank))
comm_destroy() = check(ccall((:jh_comm_destroy, LIB), Cint, ()))
allreduce_sum!(x::DevVec) = (check(ccall((:jh_comm_allreduce_sum, LIB), Cint, (Ptr{Cvoid},), handle(x))); x)
function allreduce_scalars!(vals::Vector{Float64}, op::Symbol=:sum)
    check(ccall((:jh_comm_allreduce_scalars, LIB), Cint, (Ptr{Cdouble}, Cint, Cint), vals, length(vals), op === :sum ? 0 : (op === :max ? 1 : 2)))
    vals
end
# row partition (SURVEY.md 8e): this rank owns block rows first+1 : first+count of an nrow-row tall operator
function partition_rows(nrow::Integer, world::Integer, rank::Integer)
    base, rem = divrem(nrow, world)
    (first = rank * base + min(rank, rem), count = base + (rank < rem ? 1 : 0))
end
# adjoint of a row-partitioned tall operator: local ordered sum, then the in-place all-reduce (src/Jets.jl:1045-1053 summed
# over ranks).  In `chunks` element ranges (jh_blockop_mul_adj_range) when the host wants to hand finished ranges to its own
# communication stream while the next range is computed.
function mul_adj_partitioned!(m::HipArray{T}, A::JopLn, d_local::BlockArray{T,<:HipArray{T}}; chunks::Integer=4) where {T}
    h = tall_native(A, T)
    if chunks <= 1
        check(ccall((:jh_blockop_mul_adj, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), h, handle(m), handle(d_local)))
        return allreduce_sum!(m)
    end
    # pipelined: the all-reduce of a finished range runs on the communicator's own stream while the next range is computed
    n = length(m)
    step = cld(cld(n, chunks), 16384) * 16384                             # chunk bounds on 64 KiB boundaries
    for lo = 0:step:n-1
        cnt = min(step, n - lo)
        check(ccall((:jh_blockop_mul_adj_range, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64), h, handle(m), handle(d_local), lo, cnt))
        check(ccall((:jh_comm_allreduce_sum_range, LIB), Cint, (Ptr{Cvoid}, Int64, Int64), handle(m), lo, cnt))
    end
    check(ccall((:jh_comm_join, LIB), Cint, ()))                           # the library stream waits for the exchange; no host sync
    m
end
# one Golub-Kahan step of a row-partitioned operator, pipelined the same way; returns the GLOBAL ||u|| (the one host sync)
function bidiag_step_partitioned!(u::BlockArray{T,<:HipArray{T}}, w::HipArray{T}, A::JopLn, v::HipArray{T}, alpha::Real, beta::Real; chunks::Integer=4) where {T}
    h = tall_native(A, T)
    check(ccall((:jh_normsq_reset, LIB), Cint, ()))
    n = length(w)
    step = cld(cld(n, chunks), 16384) * 16384
    for lo = 0:step:n-1
        cnt = min(step, n - lo)
        check(ccall((:jh_blockop_bidiag_step_range, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Int64, Int64, Ptr{Cdouble}),
                    h, handle(u), handle(v), handle(w), alpha, beta, lo, cnt, C_NULL))   # NULL: its share of ||u||^2 stays on the device
        check(ccall((:jh_comm_allreduce_sum_range, LIB), Cint, (Ptr{Cvoid}, Int64, Int64), handle(w), lo, cnt))
    end
    nrm2 = Ref{Cdouble}()
    check(ccall((:jh_comm_allreduce_normsq, LIB), Cint, (Ref{Cdouble},), nrm2))
    sqrt(nrm2[])
end
# ---------------------------------------------------------------- ONE Julia session, several GPUs (SURVEY section 8e)
# A context = one device + one stream + the library's workspaces for it (include/jetship.h, Conventions).  `init(device)` makes
# the device's primary context current; arrays are allocated in the CURRENT context and every operation runs in the context of
# its arrays, so a session can hold one shard of a tall operator per GPU:
#     ctxs = [ (JetsHIP.init(dev); JetsHIP.context_current()[1]) for dev in 0:ndev-1 ]
#     JetsHIP.comm_init_all(ctxs)
#     As = [ JetsHIP.with_context(ctxs[k]) do; @blockop [JopHipDiagonal(...) for i in rows_of_member_k, j=1:1]; end for k in 1:ndev ]
#     JetsHIP.mul_adj_team!(ms, As, ds)        # ms[k]: member k's replica of the domain vector, ds[k]: its rows
context_create(device::Integer) = (c = Ref{Cint}(-1); check(ccall((:jh_context_create, LIB), Cint, (Cint, Ref{Cint}), device, c)); _inited[] = true; Int(c[]))
use_context(ctx::Integer) = check(ccall((:jh_context_use, LIB), Cint, (Cint,), ctx))
set_device(device::Integer) = check(ccall((:jh_set_device, LIB), Cint, (Cint,), device))
context_destroy(ctx::Integer) = check(ccall((:jh_context_destroy, LIB), Cint, (Cint,), ctx))
function context_current()
    c, d = Ref{Cint}(-1), Ref{Cint}(-1)
    check(ccall((:jh_context_current, LIB), Cint, (Ref{Cint}, Ref{Cint}), c, d))
    (Int(c[]), Int(d[]))
end
function context_of(x::DevVec)
    c = Ref{Cint}(-1)
    check(ccall((:jh_bvec_context, LIB), Cint, (Ptr{Cvoid}, Ref{Cint}, Ptr{Cint}), handle(x), c, C_NULL))
    Int(c[])
end
function with_context(f, ctx::Integer)
    prev = context_current()[1]
    use_context(ctx)
    try
        return f()
    finally
        use_context(prev)
    end
end
comm_init_all(ctxs::Vector{<:Integer}) = check(ccall((:jh_comm_init_all, LIB), Cint, (Cint, Ptr{Cint}), length(ctxs), Cint.(ctxs)))
# the members' all-reduces of one collective, issued between ncclGroupStart / ncclGroupEnd
function comm_group(f)
    check(ccall((:jh_comm_group_begin, LIB), Cint, ()))
    try
        f()
    finally
        check(ccall((:jh_comm_group_end, LIB), Cint, ()))
    end
end
# adjoint of a tall operator whose rows are spread over the members of a team: every member's ordered row sum range by range,
# the grouped all-reduce of a finished range on the members' exchange streams while the next range computes
function mul_adj_team!(ms::Vector{<:HipArray{T}}, As::Vector{<:JopLn}, ds::Vector{<:BlockArray{T,<:HipArray{T}}}; chunks::Integer=4) where {T}
    hs = [tall_native(A, T) for A in As]
    n = length(ms[1])
    step = cld(cld(n, max(chunks, 1)), 16384) * 16384
    for lo = 0:step:n-1
        cnt = min(step, n - lo)
        for k in eachindex(As)
            check(ccall((:jh_blockop_mul_adj_range, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64), hs[k], handle(ms[k]), handle(ds[k]), lo, cnt))
        end
        comm_group() do
            for k in eachindex(As)
                check(ccall((:jh_comm_allreduce_sum_range, LIB), Cint, (Ptr{Cvoid}, Int64, Int64), handle(ms[k]), lo, cnt))
            end
        end
    end
    for k in eachindex(As)
        use_context(context_of(ms[k]))
        check(ccall((:jh_comm_join, LIB), Cint, ()))
    end
    ms
end
# one Golub-Kahan step over the team; returns the GLOBAL ||u|| (the host adds the members' deferred accumulators)
function bidiag_step_team!(us::Vector{<:BlockArray{T,<:HipArray{T}}}, ws::Vector{<:HipArray{T}}, As::Vector{<:JopLn}, vs::Vector{<:HipArray{T}},
                           alpha::Real, beta::Real; chunks::Integer=4) where {T}
    hs = [tall_native(A, T) for A in As]
    for k in eachindex(As)
        use_context(context_of(ws[k]))
        check(ccall((:jh_normsq_reset, LIB), Cint, ()))
    end
    n = length(ws[1])
    step = cld(cld(n, max(chunks, 1)), 16384) * 16384
    for lo = 0:step:n-1
        cnt = min(step, n - lo)
        for k in eachindex(As)
            check(ccall((:jh_blockop_bidiag_step_range, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Int64, Int64, Ptr{Cdouble}),
                        hs[k], handle(us[k]), handle(vs[k]), handle(ws[k]), alpha, beta, lo, cnt, C_NULL))
        end
        comm_group() do
            for k in eachindex(As)
                check(ccall((:jh_comm_allreduce_sum_range, LIB), Cint, (Ptr{Cvoid}, Int64, Int64), handle(ws[k]), lo, cnt))
            end
        end
    end
    total = 0.0
    for k in eachindex(As)
        use_context(context_of(ws[k]))
        check(ccall((:jh_comm_join, LIB), Cint, ()))
        part = Ref{Cdouble}()
        check(ccall((:jh_normsq_read, LIB), Cint, (Ref{Cdouble},), part))          # synchronises this member's stream
        total += part[]
    end
    sqrt(total)
end

# the team's operator applications, the member loop behind ONE ccall each (jh_team_mul / jh_team_mul_adj / jh_team_normal_mul):
# As[k], ds[k] = member k's block rows and its rows of the range vector, ms[k] / ys[k] = its replica of the domain vector
function _team_natives(who::AbstractString, As::Vector{<:JopLn}, ::Type{T}) where {T}
    hs = Ptr{Cvoid}[tall_native(A, T) for A in As]
    any(h -> h == C_NULL, hs) && error("$who: every member needs a device-native tall block operator")
    hs
end
function hip_team_mul!(ds::Vector{<:BlockArray{T,<:HipArray{T}}}, As::Vector{<:JopLn}, ms::Vector{<:HipArray{T}}) where {T}   # d_k = A_k m_k (src/Jets.jl:1015-1031), no exchange
    hs = _team_natives("hip_team_mul!", As, T)
    check(ccall((:jh_team_mul, LIB), Cint, (Cint, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}), length(As), hs, Ptr{Cvoid}[handle(d) for d in ds], Ptr{Cvoid}[handle(m) for m in ms]))
    ds
end
function hip_team_mul_adj!(ms::Vector{<:HipArray{T}}, As::Vector{<:JopLn}, ds::Vector{<:BlockArray{T,<:HipArray{T}}}; chunks::Integer=4) where {T}   # every m_k = sum over ALL rows (1045-1053)
    hs = _team_natives("hip_team_mul_adj!", As, T)
    check(ccall((:jh_team_mul_adj, LIB), Cint, (Cint, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Cint), length(As), hs, Ptr{Cvoid}[handle(m) for m in ms], Ptr{Cvoid}[handle(d) for d in ds], chunks))
    ms
end
function hip_team_normal_mul!(ys::Vector{<:HipArray{T}}, As::Vector{<:JopLn}, ms::Vector{<:HipArray{T}}; chunks::Integer=4) where {T}   # every y_k = (A'A) m, fused (530-534)
    hs = _team_natives("hip_team_normal_mul!", As, T)
    check(ccall((:jh_team_normal_mul, LIB), Cint, (Cint, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Cint), length(As), hs, Ptr{Cvoid}[handle(y) for y in ys], Ptr{Cvoid}[handle(m) for m in ms], chunks))
    ys
end

# the whole LSQR solve over the team behind one call: bs[k] = member k's rows of b (overwritten), xs[k] its replica of x
function hip_lsqr_team!(xs::Vector{<:HipArray{T}}, As::Vector{<:JopLn}, bs::Vector{<:BlockArray{T,<:HipArray{T}}}; x0::Bool=false, damp=0.0, atol=1e-6,
                        btol=1e-6, conlim=1e8, maxiter=100) where {T}
    hs = Ptr{Cvoid}[tall_native(A, T) for A in As]
    any(h -> h == C_NULL, hs) && error("hip_lsqr_team!: every member needs a device-native tall block operator")
    res = Ref{jh_lsqr_result}()
    hist = Vector{Cdouble}(undef, 2 * max(maxiter, 1))
    check(ccall((:jh_lsqr_solve_team, LIB), Cint, (Cint, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Cint, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint, Ref{jh_lsqr_result}, Ptr{Cdouble}),
                length(As), hs, Ptr{Cvoid}[handle(b) for b in bs], Ptr{Cvoid}[handle(x) for x in xs], x0 ? 1 : 0, damp, atol, btol, conlim, maxiter, 0, res, hist))
    xs, res[], reshape(hist, 2, :)[:, 1:res[].itn]
end

# Which of several equally sized device vectors should hold an operator's coefficients and which its range vector?  With the SAME two 64 GiB
# slabs the tall forward runs 6 % apart between the two directions (reading one and writing the other), differently in every process, and over
# three candidates the six ordered pairs span 41.9-46.3 ms per forward + adjoint (profiles/exp_r03_swap_roles.txt).  An application decides which
# allocation holds which operand once, when it builds its data:
#     coeff, d = JetsHIP.stream_pair(R)          # two UNINITIALISED block vectors of R::JetBSpace, ordered (read side, write side)
# measured with the operator's own forward + adjoint over every ordered pair of `candidates` allocations (the Python mirror's Jets.stream_pair).
function _pair_seconds(src::BlockArray{T,<:HipArray{T}}, dst::BlockArray{T,<:HipArray{T}}, calls::Integer) where {T}
    A = JetBlock_hip_diagonals(src)
    m = zeros(domain(A))
    mt = zeros(domain(A))
    mul!(dst, A, m)
    mul!(mt, A', dst)
    synchronize()
    t0 = time_ns()
    for _ = 1:calls
        mul!(dst, A, m)
        mul!(mt, A', dst)
    end
    synchronize()
    t = (time_ns() - t0) / 1e9 / calls
    close(A)
    t
end
JetBlock_hip_diagonals(x::BlockArray{T,<:HipArray{T}}) where {T} = JopLn(JetBlock(reshape(Jop[JopHipDiagonal(x.arrays[i]) for i = 1:length(x.arrays)], length(x.arrays), 1)))
function stream_pair(R::JetBSpace{T,<:HipSpace}; calls::Integer=3, candidates::Integer=2) where {T}
    xs = [Array(R) for _ = 1:max(2, candidates)]
    (length(R) * sizeof(T) < (4 << 30) || length(R.spaces) < 2) && return xs[1], xs[2]
    best, bi, bj = Inf, 1, 2
    for i = 1:length(xs), j = 1:length(xs)
        i == j && continue
        t = _pair_seconds(xs[i], xs[j], calls)
        t < best && ((best, bi, bj) = (t, i, j))
    end
    xs[bi], xs[bj]                               # (the other candidates become garbage: their memory goes to the slab cache at the next GC)
end

# measured per-operator choices (the grid walk of the tall forward): read from one operator, set on another / in another process
function tune_get(A::JopLn, name::AbstractString)
    v = Ref{Int64}()
    check(ccall((:jh_blockop_tune_get, LIB), Cint, (Ptr{Cvoid}, Cstring, Ref{Int64}), tall_native(A, eltype(range(A))), name, v))
    v[]
end
tune_set!(A::JopLn, name::AbstractString, value::Integer) = check(ccall((:jh_blockop_tune_set, LIB), Cint, (Ptr{Cvoid}, Cstring, Int64), tall_native(A, eltype(range(A))), name, value))
# range-side reductions of a row-partitioned vector: local fp64 partial, scalar all-reduce
dot_partitioned(x::BlockArray{T,<:HipArray{T}}, y::BlockArray{T,<:HipArray{T}}) where {T<:Real} = T(allreduce_scalars!(Float64[dot(x, y)])[1])
norm_partitioned(x::BlockArray{T,<:HipArray{T}}) where {T} = float(real(T))(sqrt(allreduce_scalars!(Float64[Float64(norm(x))^2])[1]))

# ---------------------------------------------------------------- the rest of the ABI (round 5: every entry point of include/jetship.h is bound)
device_count() = (n = Ref{Cint}(0); check(ccall((:jh_device_count, LIB), Cint, (Ref{Cint},), n)); Int(n[]))
function device_info()
    name = Vector{UInt8}(undef, 256)
    tot, fr, cus = Ref{Int64}(0), Ref{Int64}(0), Ref{Cint}(0)
    check(ccall((:jh_device_info, LIB), Cint, (Ptr{UInt8}, Cint, Ref{Int64}, Ref{Int64}, Ref{Cint}), name, length(name), tot, fr, cus))
    (name = unsafe_string(pointer(name)), total_mem = tot[], free_mem = fr[], cu_count = Int(cus[]))
end
shutdown() = (check(ccall((:jh_shutdown, LIB), Cint, ())); empty!(_handles); empty!(_points); empty!(_bcast_programs); _inited[] = false; nothing)
tune_get(name::AbstractString) = (v = Ref{Int64}(0); check(ccall((:jh_tune_get, LIB), Cint, (Cstring, Ref{Int64}), name, v)); v[])
# the library's stream: hand it to another library (MPI, a torch-like runtime), or make the library enqueue on the application's stream
stream() = (p = Ref{Ptr{Cvoid}}(C_NULL); check(ccall((:jh_get_stream, LIB), Cint, (Ref{Ptr{Cvoid}},), p)); p[])
set_stream!(hip_stream::Ptr{Cvoid}) = check(ccall((:jh_set_stream, LIB), Cint, (Ptr{Cvoid},), hip_stream))       # C_NULL: the library's own again
function comm_info()
    n, r = Ref{Cint}(1), Ref{Cint}(0)
    check(ccall((:jh_comm_info, LIB), Cint, (Ref{Cint}, Ref{Cint}), n, r))
    (nranks = Int(n[]), rank = Int(r[]))
end
# stream-ordered timing (what tools/ and bench.py use on the Python side)
mutable struct HipEvent
    handle::Ptr{Cvoid}
    function HipEvent()
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:jh_event_create, LIB), Cint, (Ref{Ptr{Cvoid}},), h))
        e = new(h[])
        finalizer(e -> ccall((:jh_event_destroy, LIB), Cint, (Ptr{Cvoid},), e.handle), e)
        e
    end
end
record!(e::HipEvent) = (check(ccall((:jh_event_record, LIB), Cint, (Ptr{Cvoid},), e.handle)); e)
function elapsed_ms(start::HipEvent, stop::HipEvent)          # waits for `stop`
    ms = Ref{Cfloat}(0)
    check(ccall((:jh_event_elapsed_ms, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Cfloat}), start.handle, stop.handle, ms))
    Float64(ms[])
end
# page-lock an existing host array in place for the DMA of copyto! / getblock! / setblock! (pinned_array allocates one instead)
pin!(a::Array) = (check(ccall((:jh_host_register, LIB), Cint, (Ptr{Cvoid}, Csize_t), a, sizeof(a))); a)
unpin!(a::Array) = (check(ccall((:jh_host_unregister, LIB), Cint, (Ptr{Cvoid},), a)); a)
# where block `iblock` (1-based) of a slab lives: (0-based element offset, length, device pointer) -- for interop with other HIP code
function block_info(x::BlockArray{T,<:HipArray{T}}, iblock::Integer) where {T}
    o = whole(x)
    o === nothing && error("block_info: the blocks of this BlockArray do not live in one slab")
    off, len, p = Ref{Int64}(0), Ref{Int64}(0), Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:jh_bvec_block, LIB), Cint, (Ptr{Cvoid}, Int64, Ref{Int64}, Ref{Int64}, Ref{Ptr{Cvoid}}), o.handle, iblock - 1, off, len, p))
    (offset = off[], length = len[], ptr = p[])
end
# abs.(x) of a complex device vector into a real one (test/runtests.jl:546-550 `abs.(x)` on a complex BlockArray), without the JIT
function abs!(dst::DevVec{R}, x::DevVec{Complex{R}}) where {R}
    check(ccall((:jh_abs, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), handle(dst), handle(x)))
    dst
end
# dst = c1*x1 + c2*x2 + ... left to right with every coefficient's Julia TYPE (the no-JIT spelling of `a*u .+ b*v`; k <= 8): what the
# scalar stage of the unfused chains is made of (`d .= a * m`, src/Jets.jl:1159)
function lincomb!(dst::DevVec{T}, coefs::Vector{<:Number}, xs::Vector{<:DevVec{T}}) where {T}
    length(coefs) == length(xs) || throw(DimensionMismatch("lincomb!: $(length(coefs)) coefficients for $(length(xs)) vectors"))
    c = Cdouble[]
    foreach(a -> (push!(c, real(a)); push!(c, imag(a))), coefs)
    check(ccall((:jh_lincomb_typed, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Int32}, Ptr{Ptr{Cvoid}}), handle(dst), length(xs), c, Int32[_scalar_flags(a) for a in coefs], Ptr{Cvoid}[handle(x) for x in xs]))
    dst
end
# many equally shaped broadcasts in ONE launch (F(m) / point! of thousands of elementwise-nonlinear children): program k writes dsts[k]
# from its nvec operands xs[k] (flattened) and nscal scalars (re, im pairs, flattened)
function bcast_apply_many!(progs::Vector{Ptr{Cvoid}}, dsts::Vector{<:DevVec}, xs::Vector{<:DevVec}, scalars::Vector{Cdouble}=Cdouble[])
    check(ccall((:jh_bcast_apply_many, LIB), Cint, (Cint, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Cdouble}), length(progs), progs, Ptr{Cvoid}[handle(d) for d in dsts], Ptr{Cvoid}[handle(x) for x in xs], scalars))
    dsts
end
# does the library accept this expression (compiles it for gfx950, needs no GPU)?  0 = yes
bcast_check(expr::AbstractString, ::Type{T}, nvec::Integer, nscal::Integer; real_mask::Integer=0, wide_mask::Integer=0) where {T} =
    ccall((:jh_bcast_check_typed, LIB), Cint, (Cstring, Cint, Cint, Cint, Cint, Cint), expr, dtype_code(T), nvec, real_mask, nscal, wide_mask) == 0
# unload every compiled broadcast program of this session
function release_broadcasts!()
    for (_, prog) in _bcast_programs
        ccall((:jh_bcast_destroy, LIB), Cint, (Ptr{Cvoid},), prog)
    end
    empty!(_bcast_programs)
    nothing
end
# the fused A'A of a row-partitioned operator: every rank's A_k'A_k m in `chunks` element ranges, the all-reduce of a finished range under the
# next range's kernel (what jh_cgnr_solve_partitioned does per iteration)
function normal_mul_partitioned!(y::HipArray{T}, A::JopLn, m::HipArray{T}; chunks::Integer=4) where {T}
    h = tall_native(A, T)
    n = length(y)
    step = cld(cld(n, max(chunks, 1)), 16384) * 16384
    for lo = 0:step:n-1
        cnt = min(step, n - lo)
        check(ccall((:jh_blockop_normal_mul_range, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64), h, handle(y), handle(m), lo, cnt))
        check(ccall((:jh_comm_allreduce_sum_range, LIB), Cint, (Ptr{Cvoid}, Int64, Int64), handle(y), lo, cnt))
    end
    check(ccall((:jh_comm_join, LIB), Cint, ()))
    y
end

# close(A) releases the device operator (src/Jets.jl:1120-1124 cascade)
function release!(ops)
    h = pop!(_handles, ops, C_NULL)
    delete!(_points, ops)
    h == C_NULL || ccall((:jh_blockop_destroy, LIB), Cint, (Ptr{Cvoid},), h)
    nothing
end
function Base.close(j::Jet{D,R,typeof(JetBlock_f!)}) where {D<:JetAbstractSpace,R<:JetBSpace{<:Any,<:HipSpace}}
    ops = state(j).ops
    release!(ops)
    close.(ops)
    nothing
end

end # module
