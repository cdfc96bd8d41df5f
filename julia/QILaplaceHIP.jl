# QILaplaceHIP.jl -- the `ccall` layer a QILaplace.jl maintainer adds so that `apply`, `*`,
# `coefficient`, `compress!`, `canonicalize!`, `mps_to_vector`, `norm`, `signal_mps`, `signal_ztmps`
# and `rsvd` dispatch to libqilhip.so on device-resident mirror types.
#
# NOT EXECUTABLE IN THE BUILD IMAGE (no julia binary there); the executable host mirror of the same
# ABI is the Python package `qilaplace.jl_amd`.  Entry points: include/qilaplace_hip.h.
module QILaplaceHIP

using ITensors
import ITensors: apply
import Base: *, getindex, length
import LinearAlgebra: norm
import ..Mps
import ..Mps: coefficient, compress!, canonicalize!, mps_to_vector      # extended below with device methods
using ..Mps: SignalMPS, ZTMPS, _as_signal_2n, _writeback_signal_2n
using ..Mpo: SingleSiteMPO, PairedSiteMPO
using ..ApplyMPO: _as_single_site_mpo

export DeviceMPS, DeviceMPO, to_device, to_host, signal_mps_device, marginal, mps_block, apply_compress,
    compress_mpo!, build_dt_mpo_batch, build_qft_mpo_device, build_zt_qft_chain_device, apply_coefficient_sweep, apply!, rsvd_device, svd_device,
    Comm, comm_unique_id, gather_coefficients, damping_sweep, shard_items

const LIB = get(ENV, "QILHIP_LIB", "libqilhip.so")

# ---------------------------------------------------------------- status -> exception
const QIL_OK = 0
function check(status::Cint)
    status == QIL_OK && return nothing
    msg = unsafe_string(ccall((:qil_last_error, LIB), Cstring, ()))
    status in (1, 2, 3, 7) && throw(ArgumentError(msg))   # QIL_EINVAL_LENGTH / SITES / CONFIG / ARG
    status == 4 && throw(DomainError(msg))                # QIL_EDOMAIN  (mps.jl:800,820,918)
    status == 5 && throw(OutOfMemoryError())
    status == 8 && error(msg)                             # QIL_EEMPTY   (rsvd.jl:56-60)
    error("libqilhip [status $status]: $msg")
end

# ---------------------------------------------------------------- context
mutable struct Context
    h::Ptr{Cvoid}
    function Context(device::Integer=0)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:qil_context_create, LIB), Cint, (Cint, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device, C_NULL, r))
        # No finalizer: handles (DeviceMPS / DeviceMPO) carry their own finalizers and finalizer order at exit is
        # unspecified -- a context must outlive its handles.  The library also orphans live handles when a context is
        # destroyed explicitly (destroy! below), so either order is safe.
        new(r[])
    end
end
destroy!(c::Context) = (ccall((:qil_context_destroy, LIB), Cint, (Ptr{Cvoid},), c.h); c.h = C_NULL; nothing)
const DEFAULT_CTX = Ref{Union{Nothing,Context}}(nothing)
ctx() = something(DEFAULT_CTX[], (DEFAULT_CTX[] = Context(0)))

# ---------------------------------------------------------------- device mirrors
# A device MPS keeps the ITensor Index bookkeeping of the object it was made from (sites are shared
# by `apply`, apply.jl:121) and an opaque handle to the HBM-resident tensors.
mutable struct DeviceMPS{I}
    h::Ptr{Cvoid}
    sites::Vector{I}          # for ZTMPS: interleaved main_1, copy_1, ...
    paired::Bool
    ctx::Context              # keeps the context reachable for as long as the handle is (finalizer order is unspecified)
end
mutable struct DeviceMPO{I}
    h::Ptr{Cvoid}
    sites::Vector{I}
    paired::Bool
    ctx::Context
end
DeviceMPS(h::Ptr{Cvoid}, sites::Vector{I}, paired::Bool) where {I} = DeviceMPS{I}(h, sites, paired, ctx())
DeviceMPO(h::Ptr{Cvoid}, sites::Vector{I}, paired::Bool) where {I} = DeviceMPO{I}(h, sites, paired, ctx())
_free!(x::DeviceMPS) = ccall((:qil_mps_destroy, LIB), Cint, (Ptr{Cvoid},), x.h)
_free!(x::DeviceMPO) = ccall((:qil_mpo_destroy, LIB), Cint, (Ptr{Cvoid},), x.h)

# canonical boundary layout: A[alpha, s, beta] / W[a, s', s, b], column-major, explicit dim-1 edges.
# ITensor storage order varies per tensor, so permute to the canonical index order before crossing.
function _dense_site(T::ITensor, left, phys, right)
    inds_ = Index[]
    left === nothing || push!(inds_, left)
    append!(inds_, phys)
    right === nothing || push!(inds_, right)
    A = Array(T, inds_...)
    dl = left === nothing ? 1 : dim(left)
    dr = right === nothing ? 1 : dim(right)
    return reshape(A, dl, (dim.(phys))..., dr)
end

_code(::Type{<:Real}) = Cint(0)
_code(::Type{<:Complex}) = Cint(1)
# Index identity -> the ABI's Int64 site label: the low 63 bits of the Index hash (always non-negative; no
# UInt64 / Int64 mixing in the arithmetic)
_site_id(s)::Int64 = reinterpret(Int64, hash(s) & 0x7fffffffffffffff)
_site_ids(sites) = Int64[_site_id(s) for s in sites]

function to_device(psi::SignalMPS; paired::Bool=false)
    n = length(psi.data)
    T = promote_type(map(eltype, psi.data)...)
    host = [Array{T}(_dense_site(psi.data[i], i == 1 ? nothing : psi.bonds[i-1], (psi.sites[i],),
                                 i == n ? nothing : psi.bonds[i])) for i in 1:n]
    bonds = Int64[dim(b) for b in psi.bonds]
    ids = _site_ids(psi.sites)
    ptrs = Ptr{Cvoid}[pointer(a) for a in host]
    r = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve host check(ccall((:qil_mps_create, LIB), Cint,
        (Ptr{Cvoid}, Int64, Cint, Cint, Ptr{Int64}, Ptr{Int64}, Ptr{Ptr{Cvoid}}, Cdouble, Ref{Ptr{Cvoid}}),
        ctx().h, n, _code(T), paired, bonds, ids, ptrs, psi.amplitude, r))
    return finalizer(_free!, DeviceMPS(r[], copy(psi.sites), paired))
end
to_device(psi::ZTMPS) = to_device(_as_signal_2n(psi); paired=true)            # mps.jl:421-444

function to_device(W::SingleSiteMPO; paired::Bool=false)
    n = length(W.data)
    T = promote_type(map(eltype, W.data)...)
    host = [Array{T}(_dense_site(W.data[i], i == 1 ? nothing : W.bonds[i-1], (W.sites[i]', W.sites[i]),
                                 i == n ? nothing : W.bonds[i])) for i in 1:n]   # (a, s' = in, s = out, b)
    bonds = Int64[dim(b) for b in W.bonds]
    ids = _site_ids(W.sites)
    ptrs = Ptr{Cvoid}[pointer(a) for a in host]
    r = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve host check(ccall((:qil_mpo_create, LIB), Cint,
        (Ptr{Cvoid}, Int64, Cint, Cint, Ptr{Int64}, Ptr{Int64}, Ptr{Ptr{Cvoid}}, Ref{Ptr{Cvoid}}),
        ctx().h, n, _code(T), paired, bonds, ids, ptrs, r))
    return finalizer(_free!, DeviceMPO(r[], copy(W.sites), paired))
end
to_device(W::PairedSiteMPO) = to_device(_as_single_site_mpo(W); paired=true)  # apply.jl:16-32

# ---------------------------------------------------------------- the operator API (same names)
function apply(W::DeviceMPO, psi::DeviceMPS; kwargs...)                        # apply.jl:75, :201
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:qil_apply, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), W.h, psi.h, r))
    return finalizer(_free!, DeviceMPS(r[], psi.sites, psi.paired))
end
function apply(W1::DeviceMPO, W2::DeviceMPO; kwargs...)                        # apply.jl:124, :220
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:qil_apply_mpo_mpo, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), W1.h, W2.h, r))
    return finalizer(_free!, DeviceMPO(r[], length(W1.sites) >= length(W2.sites) ? W1.sites : W2.sites, W1.paired))
end
# W * psi into an existing result of the same bond profile and dtype (steady-state loops: no allocation)
function apply!(out::DeviceMPS, W::DeviceMPO, psi::DeviceMPS)
    check(ccall((:qil_apply_into, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), W.h, psi.h, out.h))
    return out
end
*(W::DeviceMPO, psi::DeviceMPS) = apply(W, psi)                                # apply.jl:233-236
*(W1::DeviceMPO, W2::DeviceMPO) = apply(W1, W2)

function length(psi::DeviceMPS)
    n = Ref{Int64}(0)
    check(ccall((:qil_mps_nsites, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}), psi.h, n))
    return Int(n[])
end

# coefficient(psi, cfg): every front-end of mps.jl:616-645, 680-693 funnels into one bit matrix
function coefficient(psi::DeviceMPS, bits::AbstractMatrix{<:Integer})          # nb x n, one row per query
    nb, n = size(bits)
    n == length(psi) || throw(ArgumentError("coefficient: expected $(length(psi)) entries, got $n"))
    b = Matrix{UInt8}(permutedims(bits))                                        # query-major for the ABI
    out = Vector{ComplexF64}(undef, nb)
    check(ccall((:qil_coefficient_batch, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{UInt8}, Ptr{Cvoid}),
                psi.h, nb, b, out))
    return out
end
# marginals: an entry of 2 sums that site's physical index (one chain instead of 2^m coefficient calls)
function marginal(psi::DeviceMPS, bits::AbstractMatrix{<:Integer})
    nb, n = size(bits)
    b = Matrix{UInt8}(permutedims(bits))
    out = Vector{ComplexF64}(undef, nb)
    check(ccall((:qil_coefficient_marginal_batch, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{UInt8}, Ptr{Cvoid}),
                psi.h, nb, b, out))
    return out
end
coefficient(psi::DeviceMPS, cfg::AbstractVector{<:Integer}) = coefficient(psi, reshape(collect(cfg), 1, :))[1]
coefficient(psi::DeviceMPS, cfg::Tuple{Vararg{Integer}}) = coefficient(psi, collect(cfg))
coefficient(psi::DeviceMPS, cfg::Vararg{Integer}) = coefficient(psi, collect(cfg))
coefficient(psi::DeviceMPS, s::AbstractString) = coefficient(psi, Mps._parse_config_string(s))
coefficient(psi::DeviceMPS, v::Integer) = coefficient(psi, Mps._bits_from_integer(v, length(psi)))
getindex(psi::DeviceMPS, cfg::Vararg{Integer}) = coefficient(psi, collect(cfg))

function compress!(psi::DeviceMPS; maxdim::Int=typemax(Int), tol::Float64=1e-12, sweeps::Int=1)   # mps.jl:913
    check(ccall((:qil_compress, LIB), Cint, (Ptr{Cvoid}, Int64, Cdouble, Cint), psi.h, maxdim, tol, sweeps))
    return psi
end
# zip_to_compress_mpo over a whole MPO (dt_transformer.jl:167-288), in place
function compress_mpo!(W::DeviceMPO, direction::AbstractString="down"; cutoff::Float64=1e-14, maxdim::Int=1000)
    direction in ("down", "up") || error("zip_to_compress_mpo: unknown direction '$direction'")
    check(ccall((:qil_mpo_compress, LIB), Cint, (Ptr{Cvoid}, Cint, Cdouble, Int64), W.h,
                direction == "down" ? 0 : 1, cutoff, maxdim))
    return W
end
# batches of independent chains of one context: item j gets exactly compress!(items[j]; ...) / compress_mpo!(items[j], ...),
# the chains run concurrently on the context's worker streams
function compress!(items::AbstractVector{<:DeviceMPS}; maxdim::Int=typemax(Int), tol::Float64=1e-12, sweeps::Int=1)
    hs = Ptr{Cvoid}[p.h for p in items]
    check(ccall((:qil_compress_batch, LIB), Cint, (Ptr{Ptr{Cvoid}}, Int64, Int64, Cdouble, Cint), hs, length(hs), maxdim, tol,
                sweeps))
    return items
end
function compress_mpo!(items::AbstractVector{<:DeviceMPO}, direction::AbstractString="down"; cutoff::Float64=1e-14,
                       maxdim::Int=1000)
    direction in ("down", "up") || error("zip_to_compress_mpo: unknown direction '$direction'")
    hs = Ptr{Cvoid}[W.h for W in items]
    check(ccall((:qil_mpo_compress_batch, LIB), Cint, (Ptr{Ptr{Cvoid}}, Int64, Cint, Cdouble, Int64), hs, length(hs),
                direction == "down" ? 0 : 1, cutoff, maxdim))
    return items
end
function canonicalize!(psi::DeviceMPS, direction::Symbol; center=nothing, cutoff::Float64=1e-12,
                       maxdim::Int=typemax(Int))                                                   # mps.jl:787
    direction in (:right, :left) || throw(ArgumentError("Direction must be :right or :left"))
    check(ccall((:qil_canonicalize, LIB), Cint, (Ptr{Cvoid}, Cint, Int64, Cdouble, Int64),
                psi.h, direction == :right ? 0 : 1, something(center, 0), cutoff, maxdim))
    return psi
end
function norm(psi::DeviceMPS)                                                                      # mps.jl:754
    v = Ref{Cdouble}(0)
    check(ccall((:qil_norm, LIB), Cint, (Ptr{Cvoid}, Ref{Cdouble}), psi.h, v))
    return v[]
end
function mps_to_vector(psi::DeviceMPS; reverse::Bool=false)                                        # mps.jl:716
    d = Ref{Cint}(0)
    check(ccall((:qil_mps_dtype, LIB), Cint, (Ptr{Cvoid}, Ref{Cint}), psi.h, d))
    out = d[] == 1 ? Vector{ComplexF64}(undef, 2^length(psi)) : Vector{Float64}(undef, 2^length(psi))
    check(ccall((:qil_mps_to_vector, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}), psi.h, reverse, out))
    return out
end

function signal_mps_device(x::AbstractVector{<:Number}; method::Symbol=:svd, cutoff::Real=1e-15,
                           maxdim::Int=typemax(Int), k::Int=20, p::Int=10, q::Int=0,
                           random_seed::Int=1234, mindim::Int=1, paired::Bool=false)  # SignalConverters.jl:228, :247
    method in (:svd, :rsvd) || throw(ArgumentError("tensor_to_mps: unknown method $method. Use :svd or :rsvd."))
    T = eltype(x) <: Complex ? ComplexF64 : Float64
    xs = Vector{T}(x)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    meth = method == :svd ? 0 : 1
    # (the symbol in a ccall must be a literal, hence the two branches)
    if paired
        check(ccall((:qil_signal_ztmps, LIB), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Cint, Cint, Cdouble, Int64, Int64, Int64, Cint, UInt64, Int64, Ref{Ptr{Cvoid}}),
            ctx().h, xs, length(xs), _code(T), meth, cutoff, maxdim, k, p, q, random_seed, mindim, r))
    else
        check(ccall((:qil_signal_mps, LIB), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Cint, Cint, Cdouble, Int64, Int64, Int64, Cint, UInt64, Int64, Ref{Ptr{Cvoid}}),
            ctx().h, xs, length(xs), _code(T), meth, cutoff, maxdim, k, p, q, random_seed, mindim, r))
    end
    n = max(1, round(Int, log2(length(xs))))
    sites = [Index(2; tags="site-$i") for i in 1:(paired ? 2n : n)]
    return finalizer(_free!, DeviceMPS(r[], sites, paired))
end

# several signals of one length encoded concurrently (the loop over signal kinds of scripts/benchmark/zt_full_runtime.jl:151-221)
function signal_mps_device(xs::AbstractVector{<:AbstractVector{<:Number}}; method::Symbol=:svd, cutoff::Real=1e-15,
                           maxdim::Int=typemax(Int), k::Int=20, p::Int=10, q::Int=0, random_seed::Int=1234,
                           mindim::Int=1, paired::Bool=false)
    method in (:svd, :rsvd) || throw(ArgumentError("tensor_to_mps: unknown method $method. Use :svd or :rsvd."))
    T = any(x -> eltype(x) <: Complex, xs) ? ComplexF64 : Float64
    host = [Vector{T}(x) for x in xs]
    all(x -> length(x) == length(host[1]), host) || throw(ArgumentError("signal batch: all signals must have one length"))
    ptrs = Ptr{Cvoid}[pointer(x) for x in host]
    outs = fill(Ptr{Cvoid}(C_NULL), length(host))
    meth = method == :svd ? 0 : 1
    GC.@preserve host begin
        if paired
            check(ccall((:qil_signal_ztmps_batch, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Int64, Int64, Cint, Cint, Cdouble, Int64, Int64, Int64, Cint, UInt64, Int64, Ptr{Ptr{Cvoid}}),
                ctx().h, ptrs, length(host), length(host[1]), _code(T), meth, cutoff, maxdim, k, p, q, random_seed, mindim, outs))
        else
            check(ccall((:qil_signal_mps_batch, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Int64, Int64, Cint, Cint, Cdouble, Int64, Int64, Int64, Cint, UInt64, Int64, Ptr{Ptr{Cvoid}}),
                ctx().h, ptrs, length(host), length(host[1]), _code(T), meth, cutoff, maxdim, k, p, q, random_seed, mindim, outs))
        end
    end
    n = max(1, round(Int, log2(length(host[1]))))
    return [finalizer(_free!, DeviceMPS(h, [Index(2; tags="site-$i") for i in 1:(paired ? 2n : n)], paired)) for h in outs]
end

# ---------------------------------------------------------------- back to the reference's host types
# Site tensors come back in the canonical order (left bond, s, right bond); fresh bond Indices are made
# (the reference does the same after every apply: apply.jl:105-119) and the site Indices are the shared ones.
function to_host(psi::DeviceMPS)
    n = length(psi.sites)
    dims = Vector{Int64}(undef, max(n - 1, 0))
    check(ccall((:qil_mps_bond_dims, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}), psi.h, dims))
    d = Ref{Cint}(0)
    check(ccall((:qil_mps_dtype, LIB), Cint, (Ptr{Cvoid}, Ref{Cint}), psi.h, d))
    T = d[] == 1 ? ComplexF64 : Float64
    amp = Ref{Cdouble}(0)
    check(ccall((:qil_mps_amplitude, LIB), Cint, (Ptr{Cvoid}, Ref{Cdouble}), psi.h, amp))
    bonds = [Index(Int(dims[i]); tags="bond-$i") for i in 1:(n - 1)]
    data = Vector{ITensor}(undef, n)
    for i in 1:n
        dl = i == 1 ? 1 : Int(dims[i - 1])
        dr = i == n ? 1 : Int(dims[i])
        A = Array{T}(undef, dl, 2, dr)
        check(ccall((:qil_mps_download_site, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}), psi.h, i - 1, A))
        if n == 1
            data[i] = ITensor(A[1, :, 1], psi.sites[i])
        elseif i == 1
            data[i] = ITensor(A[1, :, :], psi.sites[i], bonds[i])
        elseif i == n
            data[i] = ITensor(A[:, :, 1], bonds[i - 1], psi.sites[i])
        else
            data[i] = ITensor(A, bonds[i - 1], psi.sites[i], bonds[i])
        end
    end
    sig = SignalMPS(data, psi.sites, bonds; amplitude=amp[])                    # mps.jl:75
    return psi.paired ? _writeback_signal_2n(sig) : sig                     # mps.jl:447-472
end

# to_host(::DeviceMPO): the tensors come back as W[a, s', s, b] (s' = primed = input leg); fresh bond Indices, shared
# site Indices -- the reference's constructors re-validate the result (check_singlesitempo / check_pairedsitempo,
# src/mpo.jl:30-43, 62-73).  A paired chain is returned as PairedSiteMPO (mpo.jl:62-73; the inverse of
# _as_single_site_mpo, apply.jl:16-32): even chain positions are main sites, odd ones copy sites.
function to_host(W::DeviceMPO)
    n = length(W.sites)
    dims = Vector{Int64}(undef, max(n - 1, 0))
    check(ccall((:qil_mpo_bond_dims, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}), W.h, dims))
    d = Ref{Cint}(0)
    check(ccall((:qil_mpo_dtype, LIB), Cint, (Ptr{Cvoid}, Ref{Cint}), W.h, d))
    T = d[] == 1 ? ComplexF64 : Float64
    bonds = [Index(Int(dims[i]); tags="bond-$i") for i in 1:(n - 1)]
    data = Vector{ITensor}(undef, n)
    for i in 1:n
        dl = i == 1 ? 1 : Int(dims[i - 1])
        dr = i == n ? 1 : Int(dims[i])
        A = Array{T}(undef, dl, 2, 2, dr)
        check(ccall((:qil_mpo_download_site, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}), W.h, i - 1, A))
        s = W.sites[i]
        if n == 1
            data[i] = ITensor(A[1, :, :, 1], s', s)
        elseif i == 1
            data[i] = ITensor(A[1, :, :, :], s', s, bonds[i])
        elseif i == n
            data[i] = ITensor(A[:, :, :, 1], bonds[i - 1], s', s)
        else
            data[i] = ITensor(A, bonds[i - 1], s', s, bonds[i])
        end
    end
    W.paired || return SingleSiteMPO(data, W.sites, bonds)                   # mpo.jl:30-43
    return PairedSiteMPO(data, W.sites[1:2:end], W.sites[2:2:end], bonds[2:2:end], bonds[1:2:end])   # mpo.jl:62-73
end

# ---------------------------------------------------------------- rsvd / svd on a matricised ITensor (E3)
# rsvd(A, Linds...; k, p, q, random_seed, cutoff, maxdim, mindim) (src/linalg/rsvd.jl:38-121) with the matrix work on
# the device: A is matricised over (Linds | rest) exactly as the reference does with its combiners (:62-68), the
# factors come back as ITensors U (Linds..., u), S (u, v) diagonal, V (rest..., v) -- the reference's return triple.
function _matricise(A::ITensor, Linds)
    Lis = commoninds(A, IndexSet(Linds...))
    Ris = uniqueinds(A, Lis)
    (length(Lis) == 0 || length(Ris) == 0) &&
        error("In `rsvd`, left or right index set is empty. Left inds: $(Lis), right inds: $(Ris).")   # rsvd.jl:56-60
    T = eltype(A) <: Complex ? ComplexF64 : Float64
    M = Array{T}(reshape(Array(A, Lis..., Ris...), prod(dim.(Lis)), prod(dim.(Ris))))
    return M, Lis, Ris, T
end
function _factors_to_itensors(U, S, Vh, r, Lis, Ris, bondtag)
    u = Index(r; tags=bondtag)
    v = Index(r; tags=bondtag)
    Ut = ITensor(reshape(U[:, 1:r], dim.(Lis)..., r), Lis..., u)
    St = diag_itensor(S[1:r], u, v)
    # ITensors.svd returns the triple with A ≈ U * S * V and NO dag on V (rsvd.jl:103-121 hands that triple on), so the
    # tensor V[ris, v] holds the entries of V^H: V[ris, v] = Vh[v, ris] -- a transpose, no conjugation (for a complex A a
    # conj here would break norm(A - U*S*V) ≈ 0; where Julia is available: A = random_itensor(ComplexF64, i, j);
    # U, S, V = svd_device(A, i); @assert norm(A - U * S * V) < 1e-12 * norm(A))
    Vt = ITensor(reshape(permutedims(Vh[1:r, :]), dim.(Ris)..., r), Ris..., v)
    return Ut, St, Vt
end
function rsvd_device(A::ITensor, Linds...; k::Int=20, p::Int=10, q::Int=0, random_seed::Int=1234,
                     bondtag="Link,rsvd", cutoff::Float64=1e-15, maxdim::Int=k, mindim::Int=1)
    M, Lis, Ris, T = _matricise(A, Linds)
    m, n = size(M)
    l = min(k + p, m, n)                                                       # rsvd.jl:71
    U = Matrix{T}(undef, m, l); S = Vector{Float64}(undef, l); Vh = Matrix{T}(undef, l, n)
    r = Ref{Int64}(0)
    check(ccall((:qil_rsvd, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Cint, Int64, Int64, Cint, UInt64, Cdouble, Int64, Int64, Ref{Int64},
         Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cvoid}),
        ctx().h, M, m, n, _code(T), k, p, q, random_seed, cutoff, maxdim, mindim, r, U, S, Vh))
    # the C side packs the factors for the kept rank: U m x r, Vh r x n
    rr = Int(r[])
    return _factors_to_itensors(reshape(view(vec(U), 1:(m * rr)), m, rr), S, reshape(view(vec(Vh), 1:(rr * n)), rr, n), rr,
                                Lis, Ris, bondtag)
end
# svd(A, Linds...; cutoff, maxdim, mindim) with the ITensors truncation rule (the call sites mps.jl:929,946;
# SignalConverters.jl:84,266) on the device
function svd_device(A::ITensor, Linds...; cutoff::Float64=0.0, maxdim::Int=typemax(Int), mindim::Int=1,
                    bondtag="Link,svd")
    M, Lis, Ris, T = _matricise(A, Linds)
    m, n = size(M)
    l = min(m, n)
    U = Matrix{T}(undef, m, l); S = Vector{Float64}(undef, l); Vh = Matrix{T}(undef, l, n)
    r = Ref{Int64}(0)
    check(ccall((:qil_svd_trunc, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Cint, Cdouble, Int64, Int64, Ref{Int64}, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cvoid}),
        ctx().h, M, m, n, _code(T), cutoff, maxdim, mindim, r, U, S, Vh))
    rr = Int(r[])
    return _factors_to_itensors(reshape(view(vec(U), 1:(m * rr)), m, rr), S, reshape(view(vec(Vh), 1:(rr * n)), rr, n), rr,
                                Lis, Ris, bondtag)
end

# ---------------------------------------------------------------- beyond the reference's surface (SURVEY 8f)
# compress!(apply(W, psi); maxdim, tol, sweeps) without materialising the (D chi)^2 product
function apply_compress(W::DeviceMPO, psi::DeviceMPS; maxdim::Int=typemax(Int), tol::Float64=1e-12,
                        sweeps::Int=1, zip_maxdim::Int=0)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:qil_apply_compress, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Cdouble, Cint, Int64, Ref{Ptr{Cvoid}}),
                W.h, psi.h, maxdim == typemax(Int) ? 0 : maxdim, tol, sweeps, zip_maxdim, r))
    return finalizer(_free!, DeviceMPS(r[], copy(psi.sites), psi.paired))
end
# the same for independent (operator, state) pairs -- the (signal, damping value) items of a sweep -- run concurrently
function apply_compress(Ws::AbstractVector{<:DeviceMPO}, psis::AbstractVector{<:DeviceMPS}; maxdim::Int=typemax(Int),
                        tol::Float64=1e-12, sweeps::Int=1, zip_maxdim::Int=0)
    length(Ws) == length(psis) || throw(ArgumentError("apply_compress: $(length(Ws)) operators for $(length(psis)) states"))
    hw = Ptr{Cvoid}[W.h for W in Ws]
    hp = Ptr{Cvoid}[p.h for p in psis]
    outs = fill(Ptr{Cvoid}(C_NULL), length(psis))
    check(ccall((:qil_apply_compress_batch, LIB), Cint,
                (Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Int64, Int64, Cdouble, Cint, Int64, Ptr{Ptr{Cvoid}}),
                hw, hp, length(psis), maxdim == typemax(Int) ? 0 : maxdim, tol, sweeps, zip_maxdim, outs))
    return [finalizer(_free!, DeviceMPS(h, copy(p.sites), p.paired)) for (h, p) in zip(outs, psis)]
end
# coefficient(apply(W, psi), cfg) for a batch of configurations (rows of `bits`) without forming W * psi
function coefficient(W::DeviceMPO, psi::DeviceMPS, bits::AbstractMatrix{<:Integer})
    nb, L = size(bits)
    L == length(psi.sites) || throw(ArgumentError("Configuration length $L does not match number of sites $(length(psi.sites))"))
    b = Matrix{UInt8}(permutedims(bits))                                        # site-major rows for the C side
    out = Vector{ComplexF64}(undef, nb)
    check(ccall((:qil_apply_coefficient_batch, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{UInt8}, Ptr{Cvoid}),
                W.h, psi.h, nb, b, out))
    return out
end
# all coefficients of the configurations that agree with `spec` (0 / 1 = fixed bit, 2 = summed, 3 = free) in one
# dense contraction; free sites in chain order, first one most significant (reverse = false)
function mps_block(psi::DeviceMPS, spec::AbstractVector{<:Integer}; reverse::Bool=false)
    length(spec) == length(psi.sites) ||
        throw(ArgumentError("Configuration length $(length(spec)) does not match number of sites $(length(psi.sites))"))
    d = Ref{Cint}(0)
    check(ccall((:qil_mps_dtype, LIB), Cint, (Ptr{Cvoid}, Ref{Cint}), psi.h, d))
    nfree = count(==(3), spec)
    out = d[] == 1 ? Vector{ComplexF64}(undef, 2^nfree) : Vector{Float64}(undef, 2^nfree)
    check(ccall((:qil_mps_block, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint, Ptr{Cvoid}),
                psi.h, Vector{UInt8}(spec), reverse, out))
    return out
end
# the body of a damping sweep: for every operator materialise W * psi and read it out at the same configurations
# (the loop of docs/src/tutorials/dt.jl:150-197) -- one upload, one download, one synchronisation
function apply_coefficient_sweep(Ws::Vector{<:DeviceMPO}, psi::DeviceMPS, bits::AbstractMatrix{<:Integer})
    nb = size(bits, 1)
    b = permutedims(UInt8.(bits))                       # query-major, site fastest
    out = Matrix{ComplexF64}(undef, nb, length(Ws))
    hs = Ptr{Cvoid}[W.h for W in Ws]
    check(ccall((:qil_apply_coefficient_sweep, LIB), Cint, (Ptr{Ptr{Cvoid}}, Int64, Ptr{Cvoid}, Int64, Ptr{UInt8}, Ptr{Cvoid}),
                hs, length(Ws), psi.h, nb, b, out))
    return permutedims(out)                             # (operator, query)
end
# build_dt_mpo for a sweep of damping values, built together on the device (dt_transformer.jl:312-412)
function build_dt_mpo_batch(psi::DeviceMPS, wrs::AbstractVector{<:Real}; cutoff::Float64=1e-14, maxdim::Int=1000)
    psi.paired || throw(ArgumentError("build_dt_mpo: needs a paired-register (ZTMPS) operand"))
    n = length(psi.sites) ÷ 2
    w = Vector{Float64}(wrs)
    hs = Vector{Ptr{Cvoid}}(undef, length(w))
    check(ccall((:qil_build_dt_mpo_batch, LIB), Cint,
                (Ptr{Cvoid}, Int64, Int64, Ptr{Cdouble}, Cdouble, Int64, Ptr{Int64}, Ptr{Ptr{Cvoid}}),
                ctx().h, n, length(w), w, cutoff, maxdim, _site_ids(psi.sites), hs))
    return [finalizer(_free!, DeviceMPO(h, copy(psi.sites), true)) for h in hs]
end
# build_zt_mpo(psi::ZTMPS, wr; cutoff, maxdim) (zt_transformer.jl:41-112) for a sweep of damping values, every step on the device
# behind ONE verb: the DT halves (:74) and the paired QFT chain (:78-99, built once) concurrently on two streams of the context,
# then per value apply(W_dt, mpo_qft) (:103) and zip_to_compress_mpo "down" (:104) as one batch
function build_zt_mpo_batch(psi::DeviceMPS, wrs::AbstractVector{<:Real}; cutoff::Float64=1e-14, maxdim::Int=1000)
    psi.paired || throw(ArgumentError("build_zt_mpo: needs a paired-register (ZTMPS) operand"))
    n = length(psi.sites) ÷ 2
    w = Vector{Float64}(wrs)
    hs = Vector{Ptr{Cvoid}}(undef, length(w))
    check(ccall((:qil_build_zt_mpo_batch, LIB), Cint,
                (Ptr{Cvoid}, Int64, Int64, Ptr{Cdouble}, Cdouble, Int64, Ptr{Int64}, Ptr{Ptr{Cvoid}}),
                ctx().h, n, length(w), w, cutoff, maxdim, _site_ids(psi.sites), hs))
    return [finalizer(_free!, DeviceMPO(h, copy(psi.sites), true)) for h in hs]
end
build_zt_mpo_device(psi::DeviceMPS, wr::Real; kwargs...) = build_zt_mpo_batch(psi, [wr]; kwargs...)[1]
# build_qft_mpo(psi::SignalMPS; cutoff, maxdim) (qft_transformer.jl:121-165) entirely on the device: one launch of the
# persistent complex chain builder; falls back (fallback flag) only if a bond left its in-LDS capacity, in which case the
# reference's own host builder is the route to take (build_qft_mpo(n, sites) of QILaplace.jl, then to_device).
function build_qft_mpo_device(psi::DeviceMPS; cutoff::Float64=1e-14, maxdim::Int=1000)
    psi.paired && throw(ArgumentError("build_qft_mpo: needs a single-register (SignalMPS) operand"))
    n = length(psi.sites)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    fb = Ref{Cint}(0)
    check(ccall((:qil_build_qft_mpo, LIB), Cint, (Ptr{Cvoid}, Int64, Cdouble, Int64, Ptr{Int64}, Ptr{Ptr{Cvoid}}, Ref{Cint}),
                ctx().h, n, cutoff, maxdim, _site_ids(psi.sites), h, fb))
    fb[] != 0 && error("build_qft_mpo_device: a bond exceeded the in-LDS capacity of the persistent builder; build on the host and to_device it")
    return finalizer(_free!, DeviceMPO(h[], copy(psi.sites), false))
end
# the paired-register QFT half of build_zt_mpo (zt_transformer.jl:78-99) on the device, labelled with psi's 2n sites
function build_zt_qft_chain_device(psi::DeviceMPS; cutoff::Float64=1e-14, maxdim::Int=1000)
    psi.paired || throw(ArgumentError("build_zt_mpo: needs a paired-register (ZTMPS) operand"))
    n = length(psi.sites) ÷ 2
    h = Ref{Ptr{Cvoid}}(C_NULL)
    fb = Ref{Cint}(0)
    check(ccall((:qil_build_zt_qft_chain, LIB), Cint, (Ptr{Cvoid}, Int64, Cdouble, Int64, Ptr{Int64}, Ptr{Ptr{Cvoid}}, Ref{Cint}),
                ctx().h, n, cutoff, maxdim, _site_ids(psi.sites), h, fb))
    fb[] != 0 && error("build_zt_qft_chain_device: a bond exceeded the in-LDS capacity of the persistent builder")
    return finalizer(_free!, DeviceMPO(h[], copy(psi.sites), true))
end

# ---- multi-GPU: the batched gather of a sweep (SURVEY 8e).  One Julia process per GPU (Distributed.jl workers, MPI ranks or a
# plain launcher that sets RANK / WORLD_SIZE); items are dealt round-robin (item i -> rank i mod world, `shard_items`), nothing is
# exchanged until every rank needs all coefficient batches: ONE ncclAllGather over xGMI inside qil_gather_coefficients.
# Replaces the serial loops over the damping values of docs/src/tutorials/zt.jl:300-348 / scripts/benchmark/zt_full_runtime.jl:151-221.
mutable struct Comm
    h::Ptr{Cvoid}
    rank::Int
    world::Int
    context::Any                # the Context the communicator was created on: kept alive for as long as the communicator is
end
const COMM_ID_BYTES = 128
# rank 0 creates the id and ships it to the other ranks over any host channel (Distributed.remotecall, a file, MPI.Bcast)
function comm_unique_id()
    id = Vector{UInt8}(undef, COMM_ID_BYTES)
    check(ccall((:qil_comm_unique_id, LIB), Cint, (Ptr{Cvoid},), id))
    return id
end
# collective: every rank calls it with the same id
function Comm(rank::Integer, world::Integer, id::Vector{UInt8})
    length(id) == COMM_ID_BYTES || throw(ArgumentError("Comm: the unique id must be $COMM_ID_BYTES bytes"))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    c = ctx()
    check(ccall((:qil_comm_create, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), c.h, rank, world, id, h))
    # (the library also tolerates the other order: a context destroyed first leaves the communicator's handle valid and empty)
    return finalizer(cm -> (ccall((:qil_comm_destroy, LIB), Cint, (Ptr{Cvoid},), cm.h); cm.h = C_NULL), Comm(h[], rank, world, c))
end
shard_items(n_items::Integer, world::Integer, rank::Integer) = collect(rank:world:(n_items - 1))      # zero-based item indices
# `local_batches`: this rank's coefficient batches in its own order (items rank, rank + world, ...), each of length `width`;
# returns the (n_items, width) matrix in item order on every rank
function gather_coefficients(comm::Comm, local_batches::Vector{Vector{ComplexF64}}, n_items::Integer, width::Integer)
    mine = length(shard_items(n_items, comm.world, comm.rank))
    length(local_batches) == mine || throw(ArgumentError("gather_coefficients: rank $(comm.rank) owns $mine items, got $(length(local_batches))"))
    loc = Matrix{ComplexF64}(undef, width, max(mine, 1))             # column = one item: item-major in memory
    for (s, v) in enumerate(local_batches)
        length(v) == width || throw(ArgumentError("gather_coefficients: batch $s has length $(length(v)), expected $width"))
        loc[:, s] = v
    end
    out = Matrix{ComplexF64}(undef, width, n_items)
    # (interleaved doubles on the C side; Ptr{Cvoid} because a Matrix{ComplexF64} does not convert to Ptr{Cdouble})
    check(ccall((:qil_gather_coefficients, LIB), Cint, (Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
                comm.h, n_items, width, loc, out))
    return permutedims(out)
end
# the same gather for samples that already live in HBM of the communicator's context (raw device pointers, e.g. from
# AMDGPU.jl arrays): stream-ordered, nothing crosses PCIe
function gather_coefficients_device!(comm::Comm, out_dev::Ptr{Cvoid}, local_dev::Ptr{Cvoid}, n_items::Integer, width::Integer)
    check(ccall((:qil_gather_coefficients_device, LIB), Cint, (Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
                comm.h, n_items, width, local_dev, out_dev))
    return out_dev
end
# the gather's layout rule on device buffers (world blocks of ceil(n_items / world) x width in rank order -> item order)
function sweep_unshuffle_device!(out_dev::Ptr{Cvoid}, gathered_dev::Ptr{Cvoid}, world::Integer, n_items::Integer, width::Integer)
    check(ccall((:qil_sweep_unshuffle_device, LIB), Cint, (Ptr{Cvoid}, Cint, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
                ctx().h, world, n_items, width, gathered_dev, out_dev))
    return out_dev
end
# the sweep body across the ranks: Ws = THIS rank's round-robin share of n_items operators; the samples stay in HBM, one
# all-gather, the (n_items, query) table on every rank (the loops of docs/src/tutorials/dt.jl:150-197, zt.jl:300-348)
function apply_coefficient_sweep(comm::Comm, Ws::Vector{<:DeviceMPO}, psi::DeviceMPS, bits::AbstractMatrix{<:Integer}, n_items::Integer)
    nb = size(bits, 1)
    b = permutedims(UInt8.(bits))                       # query-major, site fastest
    out = Matrix{ComplexF64}(undef, nb, n_items)
    hs = Ptr{Cvoid}[W.h for W in Ws]
    check(ccall((:qil_apply_coefficient_sweep_gather, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Int64, Ptr{Cvoid}, Int64, Ptr{UInt8}, Int64, Ptr{Cvoid}),
                comm.h, hs, length(Ws), psi.h, nb, b, n_items, out))
    return permutedims(out)                             # (item, query)
end
# the whole damping sweep of BASELINE configs[3] on this rank's share + the gather (qilaplace.jl_amd/sweep.py: damping_sweep)
function damping_sweep(comm::Comm, psi::DeviceMPS, wrs::AbstractVector{<:Real}, bits::AbstractMatrix{<:Integer}; cutoff::Float64=1e-14, maxdim::Int=1000)
    mine = shard_items(length(wrs), comm.world, comm.rank) .+ 1
    Ws = isempty(mine) ? DeviceMPO[] : build_dt_mpo_batch(psi, wrs[mine]; cutoff=cutoff, maxdim=maxdim)
    return apply_coefficient_sweep(comm, Ws, psi, bits, length(wrs))
end

end # module
