# PDMPFluxHIP.jl -- the reference-side binding a PDMPFlux.jl maintainer would add (see INTEGRATION.md).
# NOT executed anywhere in this pipeline (no Julia toolchain in the image); the same ABI is exercised
# through ctypes by tests/ and bench.py.  Layouts mirror include/pdmp_hip.h exactly.
module PDMPFluxHIP

using PDMPFlux: AbstractPDMP, PDMPHistory
import PDMPFlux: sample_skeleton, sample_from_skeleton

const LIB = get(ENV, "PDMPHIP_LIB", "libpdmphip.so")

# pdmp_config_t (include/pdmp_hip.h)
struct Config
    sampler::Cint; dim::Cint; grid_size::Cint; vectorized_bound::Cint; signed_bound::Cint
    adaptive::Cint; gaussian_velocity::Cint; deriv_mode::Cint; target_id::Cint; device::Cint
    tmax::Cdouble; refresh_rate::Cdouble
    target_params::Ptr{Cdouble}; n_target_params::Int64; max_iter_per_event::Int64
    fe_mix_p::Cdouble; fe_ran_p::Cint; fe_switch::Cint; fe_positive::Cint; fe_reserved::Cint   # ForwardECMC keywords (ABI 3)
    rh_step_size::Cdouble; rh_phi::Cdouble                                                      # RHMC keywords (ABI 4)
    fe_speed_factor::Cdouble                                                                    # ForwardECMC speed_factor (ABI 5)
end
struct Diag   # pdmp_diag_t
    horizon::Ptr{Cdouble}; ar::Ptr{Cdouble}; errored_bound::Ptr{Int32}
    error_value_ar::Ptr{Cdouble}; rejected::Ptr{Int32}; hitting_horizon::Ptr{Int32}
    is_active::Ptr{UInt8}   # StickyZigZag (ABI 5): d x n_sk x C bytes -> PDMPHistory.is_active; C_NULL otherwise
end

# target descriptors replacing the closure field `∇U` (ZigZagSamplers.jl:37)
abstract type Target end
struct GaussianIso <: Target end
struct GaussianDiag <: Target; mu::Vector{Float64}; prec::Vector{Float64}; end
struct Banana <: Target end
target_id(::GaussianIso) = 0; target_id(::GaussianDiag) = 1; target_id(::Banana) = 2
params(::Target) = Float64[]; params(t::GaussianDiag) = vcat(t.mu, t.prec)
# an arbitrary `U::Function` / `∇U` (ADBackend.jl:30-142) crosses the ABI as HIP source text, compiled by the library:
#   HIPSource("template <class X> __device__ auto U(const X &x, int d) { ... }")                 scalar potential
#   HIPGradient("template <class X> __device__ auto grad_i(int i, const X &x, int d) { ... }")   coordinate i of ∇U
struct HIPSource <: Target; text::String; end
struct HIPGradient <: Target; text::String; end
target_id(::Union{HIPSource,HIPGradient}) = 4
source_kind(::HIPSource) = 0; source_kind(::HIPGradient) = 1

# pdmp_create or, for a source target, pdmp_create_from_source
function create(cfg::Config, target::Target)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    if target isa Union{HIPSource,HIPGradient}
        check(ccall((:pdmp_create_from_source, LIB), Cint, (Ref{Config}, Cstring, Int32, Ref{Ptr{Cvoid}}),
                    cfg, target.text, source_kind(target), h))
    else
        check(ccall((:pdmp_create, LIB), Cint, (Ref{Config}, Ref{Ptr{Cvoid}}), cfg, h))
    end
    return h[]
end

mutable struct HIPSampler <: AbstractPDMP
    handle::Ptr{Cvoid}; dim::Int; sampler::Int
end

function check(rc)
    rc == 0 && return
    msg = unsafe_string(ccall((:pdmp_last_error, LIB), Cstring, ()))
    rc == -1 && throw(ArgumentError(msg))          # sample.jl:262-264, ZigZagSamplers.jl:62-68
    rc == -2 && throw(DimensionMismatch(msg))      # AbstractPDMP.jl:96-98
    error("pdmphip[$rc]: $msg")
end

# ZigZagAD(dim, target; ...) with the keyword defaults of ZigZagSamplers.jl:118-119; BPS / Boomerang alike.
function ZigZagAD(dim::Int, target::Target; refresh_rate=0.0, grid_size=10, tmax=2.0, vectorized_bound=true,
                  signed_bound=true, adaptive=true, AD_backend="ForwardDiff", device=-1)
    p = params(target)
    cfg = Config(0, dim, grid_size, vectorized_bound, signed_bound, adaptive, 0, AD_backend == "FiniteDiff" ? 1 : 0,
                 target_id(target), device, Float64(tmax), refresh_rate, pointer(p), length(p), 0, 0.5, 0, 1, 1, 0, 0.05, pi / 2, 1.0)
    handle = GC.@preserve p create(cfg, target)
    s = HIPSampler(handle, dim, 0)
    finalizer(x -> ccall((:pdmp_destroy, LIB), Cint, (Ptr{Cvoid},), x.handle), s)
    return s
end

# ForwardECMC(dim, target; ran_p, mix_p, switch, positive, ...) with the defaults of ForwardEventChainMonteCarlo.jl:290-292
function ForwardECMC(dim::Int, target::Target; grid_size=10, tmax=2.0, signed_bound=true, adaptive=true, ran_p=false,
                     mix_p=0.5, switch=true, positive=true, AD_backend="ForwardDiff", device=-1)
    p = params(target)
    cfg = Config(3, dim, grid_size, 0, signed_bound, adaptive, 0, AD_backend == "FiniteDiff" ? 1 : 0,
                 target_id(target), device, Float64(tmax), 0.0, pointer(p), length(p), 0, mix_p, ran_p, switch, positive, 0, 0.05, pi / 2, 1.0)
    handle = GC.@preserve p create(cfg, target)
    s = HIPSampler(handle, dim, 3)
    finalizer(x -> ccall((:pdmp_destroy, LIB), Cint, (Ptr{Cvoid},), x.handle), s)
    return s
end

# SpeedUpZigZag(dim, target; ...) with the defaults of SpeedUpZigZagSamplers.jl:56-59 (sampler id 4)
function SpeedUpZigZag(dim::Int, target::Target; grid_size=10, tmax=2.0, refresh_rate=0.0, vectorized_bound=true,
                       signed_bound=true, adaptive=true, AD_backend="FiniteDiff", device=-1)
    p = params(target)
    cfg = Config(4, dim, grid_size, vectorized_bound, signed_bound, adaptive, 0, AD_backend == "FiniteDiff" ? 1 : 0,
                 target_id(target), device, Float64(tmax), refresh_rate, pointer(p), length(p), 0, 0.5, 0, 1, 1, 0, 0.05, pi / 2, 1.0)
    handle = GC.@preserve p create(cfg, target)
    s = HIPSampler(handle, dim, 4)
    finalizer(x -> ccall((:pdmp_destroy, LIB), Cint, (Ptr{Cvoid},), x.handle), s)
    return s
end

# sample_skeleton(sampler, n_sk, xinit, vinit; seed) -> PDMPHistory      (src/sample.jl:253-284)
function sample_skeleton(s::HIPSampler, n_sk::Int, xinit::Vector{Float64}, vinit::Vector{Float64};
                         seed::Union{Int,Nothing}=nothing, verbose::Bool=true)
    hs = sample_skeleton(s, n_sk, reshape(xinit, :, 1), reshape(vinit, :, 1); seed=seed)
    return hs[1]
end

# ensemble method: xinit, vinit are d x C; returns one PDMPHistory per chain (column-major buffers are
# exactly the library's layout, so the histories are built from contiguous slices)
function sample_skeleton(s::HIPSampler, n_sk::Int, xinit::Matrix{Float64}, vinit::Matrix{Float64};
                         seed::Union{Int,Nothing}=nothing, chain_id_offset::Int=0)
    d, C = size(xinit)
    (d == s.dim && size(vinit) == size(xinit)) || throw(DimensionMismatch("xinit and vinit must be $(s.dim) x C"))
    n = max(n_sk, 0)
    X = Array{Float64}(undef, d, n, C); V = similar(X); t = Array{Float64}(undef, n, C)
    hz = similar(t); ar = similar(t); ev = Array{Float64}(undef, 5, n, C)
    eb = Array{Int32}(undef, n, C); rj = similar(eb); hh = similar(eb)
    sd = seed === nothing ? rand(UInt64) : reinterpret(UInt64, Int64(seed))
    GC.@preserve xinit vinit X V t hz ar ev eb rj hh begin
        diag = Diag(pointer(hz), pointer(ar), pointer(eb), pointer(ev), pointer(rj), pointer(hh), Ptr{UInt8}(C_NULL))
        check(ccall((:pdmp_sample_skeleton, LIB), Cint,
                    (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cdouble}, Ptr{Cdouble}, UInt64,
                     Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Diag}),
                    s.handle, C, n_sk, chain_id_offset, xinit, vinit, sd, X, V, t, diag))
    end
    return [PDMPHistory{Float64}(X[:, :, c], V[:, :, c], t[:, c], trues(d, n), hz[:, c], ar[:, c], eb[:, c],
                                 ev[:, :, c], rj[:, c], hh[:, c]) for c in 1:C]
end

# sample_from_skeleton(sampler, N, history; discard_vt) -> Matrix{Float64}      (src/sample.jl:475-513)
function sample_from_skeleton(s::HIPSampler, N::Int, h::PDMPHistory; discard_vt=true)
    d, n = size(h.X)
    out = Matrix{Float64}(undef, discard_vt ? d : 2d + 1, max(N, 0))
    GC.@preserve h out check(ccall((:pdmp_sample_from_skeleton, LIB), Cint,
        (Ptr{Cvoid}, Int64, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Int32, Ptr{Cdouble}),
        s.handle, 1, n, h.X, h.V, h.t, N, discard_vt, out))
    return out
end

end # module
