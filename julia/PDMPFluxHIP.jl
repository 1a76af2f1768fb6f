# PDMPFluxHIP.jl -- the reference-side binding a PDMPFlux.jl maintainer would add (see INTEGRATION.md).
# NOT executed anywhere in this pipeline (no Julia toolchain in the image); the same ABI (include/pdmp_hip.h, version 11) is
# exercised through ctypes by tests/ and bench.py.  Struct layouts mirror the header field for field.
#
# Every constructor of the reference has a method here that takes a TARGET DESCRIPTOR in place of the closure `∇U` / `U`
# (a Julia closure cannot cross a C ABI): same names, same keyword arguments, same defaults, same exceptions --
#   ZigZag / ZigZagAD          src/Samplers/ZigZagSamplers.jl:58-126
#   BPS / BPSAD                src/Samplers/BouncyParticleSamplers.jl:21-94
#   Boomerang / BoomerangAD    src/Samplers/BoomerangSamplers.jl:21-87
#   ForwardECMC / ForwardECMCAD  src/Samplers/ForwardEventChainMonteCarlo.jl:301-380
#   SpeedUpZigZag / ...AD      src/Samplers/SpeedUpZigZagSamplers.jl:58-130
#   RHMC / RHMCAD              src/Samplers/RandomizedHamiltonianMonteCarlo.jl:48-190
#   StickyZigZag / ...AD       src/Samplers/StickyZigZagSamplers.jl:61-126
# and `sample`, `sample_skeleton` (n_sk::Int and T::Float64 methods), `sample_from_skeleton` (N, dt and (N, dt) methods) dispatch on
# the returned HIPSampler exactly as src/sample.jl does on the reference samplers.
module PDMPFluxHIP

using PDMPFlux: AbstractPDMP, PDMPHistory
import PDMPFlux: sample, sample_skeleton, sample_from_skeleton

const LIB = get(ENV, "PDMPHIP_LIB", "libpdmphip.so")
const ABI_VERSION = 11

# ---------------------------------------------------------------------------------------------- C structs (include/pdmp_hip.h)
struct Config   # pdmp_config_t
    sampler::Cint; dim::Cint; grid_size::Cint; vectorized_bound::Cint; signed_bound::Cint
    adaptive::Cint; gaussian_velocity::Cint; deriv_mode::Cint; target_id::Cint; device::Cint
    tmax::Cdouble; refresh_rate::Cdouble
    target_params::Ptr{Cdouble}; n_target_params::Int64; max_iter_per_event::Int64
    fe_mix_p::Cdouble; fe_ran_p::Cint; fe_switch::Cint; fe_positive::Cint; fe_reserved::Cint   # ForwardECMC keywords
    rh_step_size::Cdouble; rh_phi::Cdouble                                                      # RHMC keywords
    fe_speed_factor::Cdouble                                                                    # ForwardECMC speed_factor
    rng_version::Cint; reserved0::Cint     # arithmetic contract of the engine's random stream and log / exp / sin / cos: 0 = default (2), 1 = rounds 1-4
end
struct Diag     # pdmp_diag_t
    horizon::Ptr{Cdouble}; ar::Ptr{Cdouble}; errored_bound::Ptr{Int32}
    error_value_ar::Ptr{Cdouble}; rejected::Ptr{Int32}; hitting_horizon::Ptr{Int32}
    is_active::Ptr{UInt8}       # StickyZigZag: d x n_sk x C bytes -> PDMPHistory.is_active; C_NULL otherwise
end
const NODIAG = Diag(C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL)

const ZIGZAG, BPS_ID, BOOMERANG, FECMC, SUZZ, RHMC_ID, STICKY = 0, 1, 2, 3, 4, 5, 6

function check(rc)
    rc == 0 && return
    msg = unsafe_string(ccall((:pdmp_last_error, LIB), Cstring, ()))
    rc == -1 && throw(ArgumentError(msg))          # sample.jl:262-264,476-478; ZigZagSamplers.jl:62-68
    rc == -2 && throw(DimensionMismatch(msg))      # AbstractPDMP.jl:96-98
    error("pdmphip[$rc]: $msg")
end
function __init__()
    v = ccall((:pdmp_abi_version, LIB), Cint, ())
    v == ABI_VERSION || error("libpdmphip.so has ABI $v, this binding expects $ABI_VERSION")
end

# ---------------------------------------------------------------------------------------------- targets
# descriptors replacing the closure field `∇U` (ZigZagSamplers.jl:37)
abstract type Target end
struct GaussianIso <: Target end                                                    # U = |x|^2 / 2 (README.md:36-38)
struct GaussianDiag <: Target; mu::Vector{Float64}; prec::Vector{Float64}; end      # U = sum prec_i (x_i - mu_i)^2 / 2
struct Banana <: Target end                                                         # U_banana, test/test_config.jl:33-36
struct GaussianPrecision <: Target; P::Matrix{Float64}; end                         # U = x'Px/2, dense precision (BASELINE config 4)
# an arbitrary `U::Function` / `∇U` (ADBackend.jl:30-142) crosses the ABI as HIP source text, compiled by the library:
#   HIPSource("template <class X> __device__ auto U(const X &x, int d) { ... }")                 scalar potential
#   HIPGradient("template <class X> __device__ auto grad_i(int i, const X &x, int d) { ... }")   coordinate i of ∇U
struct HIPSource <: Target; text::String; end
struct HIPGradient <: Target; text::String; end
target_id(::GaussianIso) = 0; target_id(::GaussianDiag) = 1; target_id(::Banana) = 2; target_id(::GaussianPrecision) = 3
target_id(::Union{HIPSource,HIPGradient}) = 4
params(::Target) = Float64[]
params(t::GaussianDiag) = vcat(t.mu, t.prec)
params(t::GaussianPrecision) = vec(copy(t.P))                                       # column-major, as the header asks
source_kind(::HIPSource) = 0; source_kind(::HIPGradient) = 1

# ---------------------------------------------------------------------------------------------- samplers
mutable struct HIPSampler <: AbstractPDMP
    handle::Ptr{Cvoid}
    dim::Int
    sampler::Int
    name::String
    state::Union{Nothing,PDMPHistory}       # `sampler.state = state` of sample.jl:281 (the last history column)
end
sticky(s::HIPSampler) = s.sampler == STICKY

deriv_mode(ad::String) = ad in ("ForwardDiff", "Zygote", "ReverseDiff", "Enzyme", "PolyesterForwardDiff") ? Cint(0) :
                         ad in ("FiniteDiff", "Undefined", "") ? Cint(1) : throw(ArgumentError("Unsupported AD_backend: $ad"))   # UpperBound.jl:120,226

function make(id::Int, name::String, dim::Int, target::Target; grid_size=10, tmax=2.0, refresh_rate=0.0, vectorized_bound=true,
              signed_bound=true, adaptive=true, AD_backend="FiniteDiff", Gaussian_velocity=false, device=-1, extra=Float64[],
              mix_p=0.5, ran_p=false, switch=true, positive=true, speed_factor=1.0, step_size=0.05, phi=pi / 2)
    dim > 0 || throw(ArgumentError("dimension dim must be positive. Current value: $dim"))             # ZigZagSamplers.jl:62-64
    grid_size >= 0 || throw(ArgumentError("grid_size must be non-negative. Current value: $grid_size"))  # :66-68
    p = vcat(params(target), extra)                                   # StickyZigZag: kappa travels as the last dim parameters
    cfg = Config(id, dim, grid_size, vectorized_bound, signed_bound, adaptive, Gaussian_velocity, deriv_mode(AD_backend),
                 target_id(target), device, Float64(tmax), Float64(refresh_rate), pointer(p), length(p), 0,
                 Float64(mix_p), ran_p, switch, positive, 0, Float64(step_size), Float64(phi), Float64(speed_factor), 0, 0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve p begin
        if target isa Union{HIPSource,HIPGradient}
            check(ccall((:pdmp_create_from_source, LIB), Cint, (Ref{Config}, Cstring, Int32, Ref{Ptr{Cvoid}}),
                        cfg, target.text, source_kind(target), h))
        else
            check(ccall((:pdmp_create, LIB), Cint, (Ref{Config}, Ref{Ptr{Cvoid}}), cfg, h))
        end
    end
    s = HIPSampler(h[], dim, id, name, nothing)
    finalizer(x -> ccall((:pdmp_destroy, LIB), Cint, (Ptr{Cvoid},), x.handle), s)
    return s
end

function warn_signed(vectorized_bound, signed_bound)                                 # ZigZagSamplers.jl:75-78
    if signed_bound && !vectorized_bound
        @warn "Signed bound is not compatible with non-vectorized bound for ZigZag, switching to unsigned bound"
        return false
    end
    return signed_bound
end

ZigZag(dim::Int, target::Target; grid_size=10, tmax=2.0, refresh_rate=0.0, vectorized_bound=true, signed_bound=true, adaptive=true,
       AD_backend="FiniteDiff", device=-1) =
    make(ZIGZAG, "ZigZag", dim, target; grid_size, tmax, refresh_rate, vectorized_bound, signed_bound=warn_signed(vectorized_bound, signed_bound),
         adaptive, AD_backend, device)
ZigZagAD(dim::Int, target::Target; refresh_rate=0.0, grid_size=10, tmax=2.0, vectorized_bound=true, signed_bound=true, adaptive=true,
         AD_backend="ForwardDiff", device=-1) =
    ZigZag(dim, target; grid_size, tmax, refresh_rate, vectorized_bound, signed_bound, adaptive, AD_backend, device)

BPS(dim::Int, target::Target; grid_size=10, tmax=1.0, refresh_rate=0.1, vectorized_bound=false, signed_bound=true, adaptive=true,
    AD_backend="ForwardDiff", Gaussian_velocity=false, device=-1) =                   # vectorized_bound forced false, :37
    make(BPS_ID, "BPS", dim, target; grid_size, tmax, refresh_rate, vectorized_bound=false, signed_bound, adaptive, AD_backend,
         Gaussian_velocity, device)
BPSAD(dim::Int, target::Target; refresh_rate=0.0, grid_size=10, tmax=2.0, vectorized_bound=true, signed_bound=true, adaptive=true,
      AD_backend="ForwardDiff", device=-1) =
    BPS(dim, target; grid_size, tmax, refresh_rate, signed_bound, adaptive, AD_backend, device)

Boomerang(dim::Int, target::Target; grid_size=10, tmax=1.0, refresh_rate=0.1, vectorized_bound=false, signed_bound=true, adaptive=true,
          AD_backend="FiniteDiff", device=-1) =
    make(BOOMERANG, "Boomerang", dim, target; grid_size, tmax, refresh_rate, vectorized_bound=false, signed_bound, adaptive, AD_backend, device)
BoomerangAD(dim::Int, target::Target; refresh_rate=0.0, grid_size=10, tmax=2.0, vectorized_bound=true, signed_bound=true, adaptive=true,
            AD_backend="ForwardDiff", device=-1) =
    Boomerang(dim, target; grid_size, tmax, refresh_rate, signed_bound, adaptive, AD_backend, device)

function ForwardECMC(dim::Int, target::Target; grid_size=10, tmax=2.0, signed_bound=true, adaptive=true, ran_p=false, mix_p=0.5,
                     switch=true, positive=true, speed_factor=1.0, normal=false, AD_backend="FiniteDiff", device=-1)
    dim >= 2 || throw(ArgumentError("The dimension must be at least 2 to use the ForwardEventChain. Got dimension $dim"))
    normal && throw(ArgumentError("normal = true: the reference's variant takes the square root of a vector (ForwardEventChainMonteCarlo.jl:228) and cannot run"))
    make(FECMC, "ForwardECMC", dim, target; grid_size, tmax, refresh_rate=0.0, vectorized_bound=false, signed_bound, adaptive, AD_backend,
         device, mix_p=(dim == 2 ? 0.0 : mix_p), ran_p, switch, positive, speed_factor)
end
ForwardECMCAD(dim::Int, target::Target; AD_backend="ForwardDiff", kw...) = ForwardECMC(dim, target; AD_backend, kw...)

SpeedUpZigZag(dim::Int, target::Target; grid_size=10, tmax=2.0, refresh_rate=0.0, vectorized_bound=true, signed_bound=true, adaptive=true,
              AD_backend="FiniteDiff", device=-1) =
    make(SUZZ, "SpeedUpZigZag", dim, target; grid_size, tmax, refresh_rate, vectorized_bound,
         signed_bound=warn_signed(vectorized_bound, signed_bound), adaptive, AD_backend, device)
SpeedUpZigZagAD(dim::Int, target::Target; AD_backend="ForwardDiff", kw...) = SpeedUpZigZag(dim, target; AD_backend, kw...)

function RHMC(dim::Int, target::Target; mean_duration=nothing, refresh_rate=1.0, phi=pi / 2, step_size=0.05, tmax=10.0, adaptive=false,
              AD_backend="FiniteDiff", device=-1)
    if mean_duration !== nothing                                                      # :63-69
        (isfinite(mean_duration) && mean_duration > 0) || throw(ArgumentError("mean_duration must be finite and positive. Current value: $mean_duration"))
        refresh_rate = 1.0 / mean_duration
    end
    (isfinite(refresh_rate) && refresh_rate > 0) || throw(ArgumentError("refresh_rate must be finite and positive. Current value: $refresh_rate"))
    (0 < phi <= pi / 2) || throw(ArgumentError("phi must satisfy 0 < phi <= pi/2. Current value: $phi"))
    (isfinite(step_size) && step_size > 0) || throw(ArgumentError("step_size must be finite and positive. Current value: $step_size"))
    (isfinite(tmax) && tmax >= 0) || throw(ArgumentError("tmax must be finite and non-negative. Current value: $tmax"))
    make(RHMC_ID, "RHMC", dim, target; grid_size=0, tmax, refresh_rate, vectorized_bound=false, signed_bound=false, adaptive, AD_backend,
         device, step_size, phi)                                                      # :91-95
end
RHMCAD(dim::Int, target::Target; AD_backend="ForwardDiff", kw...) = RHMC(dim, target; AD_backend, kw...)

function StickyZigZag(dim::Int, target::Target, κ::Vector{Float64}; refresh_rate=0.0, grid_size=10, tmax=2.0, vectorized_bound=true,
                      signed_bound=true, adaptive=true, AD_backend="FiniteDiff", device=-1)
    length(κ) == dim || throw(DimensionMismatch("kappa must have length dim ($dim)"))
    make(STICKY, "StickyZigZag", dim, target; grid_size, tmax, refresh_rate, vectorized_bound,
         signed_bound=warn_signed(vectorized_bound, signed_bound), adaptive, AD_backend, device, extra=κ)
end
StickyZigZagAD(dim::Int, target::Target, κ::Vector{Float64}; AD_backend="ForwardDiff", kw...) = StickyZigZag(dim, target, κ; AD_backend, kw...)

# ---------------------------------------------------------------------------------------------- histories
seed_word(seed) = seed === nothing ? rand(UInt64) : reinterpret(UInt64, Int64(seed))     # Random.default_rng() / MersenneTwister(seed), AbstractPDMP.jl:100

struct Buffers
    X::Array{Float64,3}; V::Array{Float64,3}; t::Matrix{Float64}
    hz::Matrix{Float64}; ar::Matrix{Float64}; ev::Array{Float64,3}
    eb::Matrix{Int32}; rj::Matrix{Int32}; hh::Matrix{Int32}; ia::Array{UInt8,3}
end
Buffers(d, n, C) = Buffers(Array{Float64}(undef, d, n, C), Array{Float64}(undef, d, n, C), Array{Float64}(undef, n, C),
                           Array{Float64}(undef, n, C), Array{Float64}(undef, n, C), Array{Float64}(undef, 5, n, C),
                           Array{Int32}(undef, n, C), Array{Int32}(undef, n, C), Array{Int32}(undef, n, C), ones(UInt8, d, n, C))
diag_of(b::Buffers, with_active::Bool) = Diag(pointer(b.hz), pointer(b.ar), pointer(b.eb), pointer(b.ev), pointer(b.rj), pointer(b.hh),
                                              with_active ? pointer(b.ia) : Ptr{UInt8}(C_NULL))
history(b::Buffers, c::Int, k::Int) = PDMPHistory{Float64}(b.X[:, 1:k, c], b.V[:, 1:k, c], b.t[1:k, c], BitMatrix(b.ia[:, 1:k, c] .!= 0),
                                                           b.hz[1:k, c], b.ar[1:k, c], b.eb[1:k, c], b.ev[:, 1:k, c], b.rj[1:k, c], b.hh[1:k, c])

function run_fixed(s::HIPSampler, n_sk::Int, xinit::Matrix{Float64}, vinit::Matrix{Float64}, seed, chain_id_offset::Int)
    d, C = size(xinit)
    (d == s.dim && size(vinit) == size(xinit)) || throw(DimensionMismatch(
        "xinit and vinit must have the same dimension as pdmp.dim ($(s.dim)). Current dimensions: xinit ($d), vinit ($(size(vinit, 1)))"))   # AbstractPDMP.jl:96-98
    n_sk > 0 || throw(ArgumentError("n_sk must be positive. Current value: $n_sk"))                   # sample.jl:262-264
    b = Buffers(d, n_sk, C)
    GC.@preserve xinit vinit b begin
        dg = diag_of(b, sticky(s))
        check(ccall((:pdmp_sample_skeleton, LIB), Cint,
                    (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cdouble}, Ptr{Cdouble}, UInt64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Diag}),
                    s.handle, C, n_sk, chain_id_offset, xinit, vinit, seed_word(seed), b.X, b.V, b.t, dg))
    end
    return b
end

# sample_skeleton(sampler, n_sk::Int, xinit, vinit; seed, verbose) -> PDMPHistory      (src/sample.jl:253-284)
function sample_skeleton(s::HIPSampler, n_sk::Int, xinit::Vector{Float64}, vinit::Vector{Float64};
                         seed::Union{Int,Nothing}=nothing, verbose::Bool=true)
    b = run_fixed(s, n_sk, reshape(xinit, :, 1), reshape(vinit, :, 1), seed, 0)
    s.state = history(b, 1, n_sk)
    return s.state
end
# 1-d dispatch (:289-311)
function sample_skeleton(s::HIPSampler, n_sk::Int, xinit::Union{Float64,Int}, vinit::Union{Float64,Int}; kw...)
    (isfinite(xinit) && isfinite(vinit)) || throw(ArgumentError("initial values contain NaN, Inf, or -Inf."))
    return sample_skeleton(s, n_sk, [Float64(xinit)], [Float64(vinit)]; kw...)
end
# ensemble method: xinit, vinit are d x C (chain c = column c; stream (seed, chain_id_offset + c)); one PDMPHistory per chain
sample_skeleton(s::HIPSampler, n_sk::Int, xinit::Matrix{Float64}, vinit::Matrix{Float64}; seed::Union{Int,Nothing}=nothing, chain_id_offset::Int=0) =
    (b = run_fixed(s, n_sk, xinit, vinit, seed, chain_id_offset); [history(b, c, n_sk) for c in 1:size(xinit, 2)])

# sample_skeleton(sampler, T::Float64, xinit, vinit; seed, verbose, init_capacity) -> PDMPHistory with t[end] == T   (src/sample.jl:323-439)
# The fixed-count kernel runs with a capacity that doubles (as _grow_history does, Composites.jl:172-191) until the chain has passed T;
# pdmp_close_at_time_device writes the closing column (the sampler's own flow, the Sticky mask, zeroed statistics); needs device buffers,
# so this method goes through AMDGPU.jl's ROCArray (any array type with a device `pointer` works).
function sample_skeleton(s::HIPSampler, T::Float64, xinit::Vector{Float64}, vinit::Vector{Float64};
                         seed::Union{Int,Nothing}=nothing, verbose::Bool=true, init_capacity::Int=1024, device_array=nothing)
    (isfinite(T) && T >= 0) || throw(ArgumentError("T must be finite and non-negative. Current value: $T"))   # :335-337
    device_array === nothing && error("pass device_array = AMDGPU.ROCArray (the time-horizon method keeps the skeleton in HBM)")
    A = device_array
    d = s.dim
    cap = max(2, init_capacity)
    sd = seed_word(seed)
    x, v = A(reshape(xinit, :, 1)), A(reshape(vinit, :, 1))
    while true
        X, V, t = A{Float64}(undef, d, cap, 1), A{Float64}(undef, d, cap, 1), A{Float64}(undef, cap, 1)
        hz, ar, ev = A{Float64}(undef, cap, 1), A{Float64}(undef, cap, 1), A{Float64}(undef, 5, cap, 1)
        eb, rj, hh = A{Int32}(undef, cap, 1), A{Int32}(undef, cap, 1), A{Int32}(undef, cap, 1)
        ia = A(ones(UInt8, d, cap, 1))
        count = A{Int64}(undef, 1)
        dg = Diag(pointer(hz), pointer(ar), pointer(eb), pointer(ev), pointer(rj), pointer(hh), pointer(ia))
        check(ccall((:pdmp_sample_skeleton_device, LIB), Cint,
                    (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, UInt64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Diag}, Ptr{Cvoid}),
                    s.handle, 1, cap, 0, pointer(x), pointer(v), sd, pointer(X), pointer(V), pointer(t), dg, C_NULL))
        check(ccall((:pdmp_close_at_time_device, LIB), Cint,
                    (Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Diag}, Cdouble, Ptr{Cvoid}, Ptr{Cvoid}),
                    s.handle, 1, cap, pointer(X), pointer(V), pointer(t), dg, T, pointer(count), C_NULL))
        check(ccall((:pdmp_check_status, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}), s.handle, C_NULL))
        k = Array(count)[1]
        if k > 0
            b = Buffers(Array(X), Array(V), Array(t), Array(hz), Array(ar), Array(ev), Array(eb), Array(rj), Array(hh), Array(ia))
            s.state = history(b, 1, Int(k))                                       # _trim_history, Composites.jl:198-214
            return s.state
        end
        cap *= 2                                                                  # newcap = 2 * size, sample.jl:375-377
    end
end

# sample_from_skeleton(sampler, N::Int, history; discard_vt)       (src/sample.jl:475-513; StickyPDMP method :516-561)
function sample_from_skeleton(s::HIPSampler, N::Int, h::PDMPHistory; discard_vt=true)
    N > 0 || throw(ArgumentError("N must be positive. Current value: $N"))
    d, n = size(h.X)
    out = Matrix{Float64}(undef, discard_vt ? d : 2d + 1, N)
    ia = Array{UInt8}(h.is_active)
    GC.@preserve h out ia check(ccall((:pdmp_sample_from_skeleton, LIB), Cint,
        (Ptr{Cvoid}, Int64, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{UInt8}, Int64, Int32, Ptr{Cdouble}),
        s.handle, 1, n, h.X, h.V, h.t, ia, N, discard_vt, out))
    return out
end
# sample_from_skeleton(sampler, dt::Float64, history; discard_vt)   (:573-646; StickyPDMP: use_active)
function sample_from_skeleton(s::HIPSampler, dt::Float64, h::PDMPHistory; discard_vt::Bool=true)
    d, n = size(h.X)
    n_skel = Int(floor(h.t[end] / dt))                                             # :608
    out = Matrix{Float64}(undef, discard_vt ? d : 2d + 1, n_skel)
    n_skel == 0 && return out
    ia = Array{UInt8}(h.is_active)
    GC.@preserve h out ia check(ccall((:pdmp_sample_from_skeleton_dt, LIB), Cint,
        (Ptr{Cvoid}, Int64, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{UInt8}, Int64, Cdouble, Int32, Ptr{Cdouble}, Ptr{Int64}),
        s.handle, 1, n, h.X, h.V, h.t, ia, n_skel, dt, discard_vt, out, C_NULL))
    return out
end
# sample_from_skeleton(sampler, N::Int, dt::Float64, history; discard_vt)   (:649-682: first N columns, tm = dt:dt:t[N], v .* is_active)
function sample_from_skeleton(s::HIPSampler, N::Int, dt::Float64, h::PDMPHistory; discard_vt=true)
    d, n = size(h.X)
    1 <= N <= n || throw(BoundsError(h.t, N))
    M = length(dt:dt:h.t[N])
    out = Matrix{Float64}(undef, discard_vt ? d : 2d + 1, M)
    M == 0 && return out
    ia = Array{UInt8}(h.is_active)
    GC.@preserve h out ia check(ccall((:pdmp_sample_from_skeleton_prefix, LIB), Cint,
        (Ptr{Cvoid}, Int64, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{UInt8}, Int64, Cdouble, Int64, Int32, Ptr{Cdouble}, Ptr{Int64}),
        s.handle, 1, n, h.X, h.V, h.t, ia, N, dt, M, discard_vt, out, C_NULL))
    return out
end

# sample(sampler, N_sk, N_samples, xinit, vinit; seed, verbose, discard_vt)      (src/sample.jl:27-58)
function sample(s::HIPSampler, N_sk::Int, N_samples::Int, xinit, vinit; seed::Union{Int,Nothing}=nothing, verbose::Bool=true, discard_vt::Bool=true)
    history = sample_skeleton(s, N_sk, xinit, vinit; seed=seed, verbose=verbose)
    return sample_from_skeleton(s, N_samples, history; discard_vt=discard_vt)
end

# ---------------------------------------------------------------------------------------------- device and multi-device entries
# HBM-resident ensembles: `xinit`, `vinit`, `X`, `V`, `t` are device arrays (AMDGPU.ROCArray); asynchronous on `stream`.
function sample_skeleton_device!(s::HIPSampler, X, V, t, xinit, vinit; seed::Int, chain_id_offset::Int=0, stream=C_NULL)
    d, n_sk, C = size(X)
    check(ccall((:pdmp_sample_skeleton_device, LIB), Cint,
                (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, UInt64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                s.handle, C, n_sk, chain_id_offset, pointer(xinit), pointer(vinit), seed_word(seed), pointer(X), pointer(V), pointer(t), C_NULL, stream))
    return nothing
end
check_status(s::HIPSampler) = check(ccall((:pdmp_check_status, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}), s.handle, C_NULL))
# which kernel form served the last launch (diagnostic; waits for the launch): 1 wave per chain, 2 group (LDS), 3 group (registers),
# 4 one lane per chain (large Zig-Zag ensembles, d = 1..52, unit speeds), 5 dense-precision path, 6 potential given as source,
# 7 a pair of lanes per chain (Zig-Zag, d = 53..104; mid-sized launches at d = 1..52)
last_launch_form(s::HIPSampler) = Int(ccall((:pdmp_last_launch_form, LIB), Cint, (Ptr{Cvoid},), s.handle))

# One Julia process, n GPUs (pdmp_multi_*): the C chains shard into contiguous ranges over the devices; with gather = true one RCCL
# collective per output array leaves the whole skeleton on every device (the final exchange of a multi-GPU run).
#   X[r], V[r], t[r]: device arrays on device r -- full-size (d x n_sk x C) with gather, shard-sized otherwise.
mutable struct MultiSampler
    handle::Ptr{Cvoid}
    n::Int
end
function MultiSampler(cfg::Config, n_devices::Int; devices::Union{Nothing,Vector{Int32}}=nothing)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:pdmp_multi_create, LIB), Cint, (Ref{Config}, Int32, Ptr{Int32}, Ref{Ptr{Cvoid}}), cfg, n_devices,
                devices === nothing ? Ptr{Int32}(C_NULL) : pointer(devices), h))
    m = MultiSampler(h[], n_devices)
    finalizer(x -> ccall((:pdmp_multi_destroy, LIB), Cint, (Ptr{Cvoid},), x.handle), m)
    return m
end
function shard_range(C::Int, rank::Int, n::Int)      # 0-based rank; the partition pdmp_multi_sample_skeleton uses
    lo, hi = Ref{Int64}(0), Ref{Int64}(0)
    check(ccall((:pdmp_multi_shard_range, LIB), Cint, (Int64, Int32, Int32, Ref{Int64}, Ref{Int64}), C, rank, n, lo, hi))
    return lo[] + 1, hi[]                            # 1-based inclusive
end
function sample_skeleton!(m::MultiSampler, X::Vector, V::Vector, t::Vector, xinit::Matrix{Float64}, vinit::Matrix{Float64}, n_sk::Int;
                          seed::Int, chain_id_offset::Int=0, gather::Bool=true)
    C = size(xinit, 2)
    px, pv, pt = [Ptr{Cvoid}(pointer(a)) for a in X], [Ptr{Cvoid}(pointer(a)) for a in V], [Ptr{Cvoid}(pointer(a)) for a in t]
    GC.@preserve xinit vinit px pv pt check(ccall((:pdmp_multi_sample_skeleton, LIB), Cint,
        (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cdouble}, Ptr{Cdouble}, UInt64, Int32, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}),
        m.handle, C, n_sk, chain_id_offset, xinit, vinit, seed_word(seed), gather, px, pv, pt))
    return nothing
end

# ---- ABI 9: device-resident initial states, chunked device memory for the output arrays, the launch policy
# Shards of the initial states already on their devices: xinit[r], vinit[r] = d x len_r device arrays on device r (no PCIe staging per call).
function sample_skeleton_device!(m::MultiSampler, X::Vector, V::Vector, t::Vector, xinit::Vector, vinit::Vector, C::Int, n_sk::Int;
                                 seed::Int, chain_id_offset::Int=0, gather::Bool=true)
    px, pv, pt = [Ptr{Cvoid}(pointer(a)) for a in X], [Ptr{Cvoid}(pointer(a)) for a in V], [Ptr{Cvoid}(pointer(a)) for a in t]
    ix, iv = [Ptr{Cvoid}(pointer(a)) for a in xinit], [Ptr{Cvoid}(pointer(a)) for a in vinit]
    GC.@preserve px pv pt ix iv check(ccall((:pdmp_multi_sample_skeleton_device, LIB), Cint,
        (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, UInt64, Int32, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}),
        m.handle, C, n_sk, chain_id_offset, ix, iv, seed_word(seed), gather, px, pv, pt))
    return nothing
end
# Synthetic initial states on the device (BASELINE.md section 3): x = x_mean + x_scale N(0, I); v_kind 0: +-1 coordinates, 1: N(0, I),
# 2: uniform on the unit sphere -- from stream 1 of each chain's Philox key (seed, chain id), the chains' own draws come from stream 0.
function init_states_device!(s::HIPSampler, xinit, vinit; seed::Int, chain_id_offset::Int=0, x_mean::Float64=0.0, x_scale::Float64=1.0,
                             v_kind::Int=0, stream=C_NULL)
    d, C = size(xinit)
    check(ccall((:pdmp_init_states_device, LIB), Cint, (Ptr{Cvoid}, Int64, Int64, UInt64, Cdouble, Cdouble, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                s.handle, C, chain_id_offset, seed_word(seed), x_mean, x_scale, v_kind, pointer(xinit), pointer(vinit), stream))
    return nothing
end
# Device memory built from independent 1-GiB physical chunks for the large output arrays (DESIGN.md section 6): wrap the pointer with
# `unsafe_wrap(ROCArray, Ptr{Float64}(p), dims)`; release with device_free (waits for the device first).
function device_alloc(bytes::Integer; device::Int=-1, chunk_bytes::Integer=0)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:pdmp_device_alloc, LIB), Cint, (Int32, Csize_t, Csize_t, Ref{Ptr{Cvoid}}), device, bytes, chunk_bytes, p)
    rc == 0 || error(unsafe_string(ccall((:pdmp_device_alloc_error, LIB), Cstring, ())))
    return p[]
end
device_free(p::Ptr{Cvoid}) = ccall((:pdmp_device_free, LIB), Cint, (Ptr{Cvoid},), p)
# X, V (d x n_sk x C) and t (n_sk x C) of an ensemble, drawn up to `draws` times: each draw a fresh set of chunks, judged by a kernel with
# the launch's store pattern, the fastest set kept (ABI 10, DESIGN.md section 6b).  Returns the three pointers (release each with
# device_free) and the time of every draw in ms.
function alloc_outputs(d::Integer, n_sk::Integer, C::Integer; device::Int=-1, draws::Integer=3)
    sizes = Csize_t[8 * d * n_sk * C, 8 * d * n_sk * C, 8 * n_sk * C]
    ptrs = fill(C_NULL, 3)
    ms = zeros(Cfloat, max(draws, 1))
    drawn = Ref{Int32}(0)
    rc = ccall((:pdmp_device_alloc_streams, LIB), Cint, (Int32, Int32, Ptr{Csize_t}, Csize_t, Csize_t, Int32, Ptr{Ptr{Cvoid}}, Ptr{Cfloat}, Ref{Int32}),
               device, 3, sizes, 0, 8 * d * n_sk, draws, ptrs, ms, drawn)
    rc == 0 || error(unsafe_string(ccall((:pdmp_device_alloc_error, LIB), Cstring, ())))
    return (X = ptrs[1], V = ptrs[2], t = ptrs[3], draw_ms = draws > 1 ? ms[1:drawn[]] : Cfloat[])
end
# The PDMP_* environment overrides of the launch policy are read once, when the handle is created; re-read them after changing ENV.
reload_policy!(s::HIPSampler) = check(ccall((:pdmp_reload_policy, LIB), Cint, (Ptr{Cvoid},), s.handle))

# ---------------------------------------------------------------------------------------------- time horizon for an ensemble (ABI 8)
# sample_skeleton(sampler, T::Float64, xinit, vinit; seed, init_capacity) (src/sample.jl:323-439) for C chains held in HBM, with the time
# horizon inside the kernel: the history comes in chunks -- `init_capacity` columns for every chain, then smaller chunks for the chains
# that have not reached T, which resume from their hand-off state; finished chains cost nothing, memory is the sum of the history
# lengths.  `A` = the device array type (AMDGPU.ROCArray); xinit, vinit: d x C device arrays.  Returns the packed `_trim_history`
# layout (Composites.jl:198-214): X, V (d x total), t (total), offsets (C + 1; chain c = columns offsets[c]+1 : offsets[c+1]), count (C).
# PDMP_ERR_UNSUPPORTED (-5: tape replays; the dense-precision target with fewer than 64 chains or d > 1024 -- the batched dense forms serve the
# horizon in-kernel since round 4) -> use the single-chain method above per chain.  StickyZigZag needs
# the PDMPHistory.is_active arrays (pdmp_diag_t.is_active, ABI 8) for its closing column (sample.jl:386-401); this method records X, V, t
# only and therefore refuses a sticky sampler instead of returning an unmasked closing column.
function sample_skeleton_until(s::HIPSampler, T::Float64, xinit, vinit, A; seed::Int, chain_id_offset::Int=0, init_capacity::Int=1024, stream=C_NULL)
    (isfinite(T) && T >= 0) || throw(ArgumentError("T must be finite and non-negative. Current value: $T"))   # :335-337
    sticky(s) && throw(ArgumentError("sample_skeleton_until: StickyZigZag histories carry is_active; use the single-chain method"))
    d, C = size(xinit)
    # the lane forms address a wave's window of X / V through a 32-bit buffer descriptor: 64 * capacity * d * 8 < 2^31 (engine_api.hip: lane_ok)
    cap_limit = d <= 104 ? 8 * fld(fld(2^31 - 1, 64 * 8 * d), 8) : typemax(Int)
    cap = min(8 * cld(max(2, init_capacity), 8), max(cap_limit, 8))
    cap_sum = 0
    total = A(zeros(Int64, C))
    chunks = NamedTuple[]
    cmap, Cl, skip = nothing, C, 0
    while true
        X, V, t = A{Float64}(undef, d, cap, Cl), A{Float64}(undef, d, cap, Cl), A{Float64}(undef, cap, Cl)
        nrec, count, start, nxt, nact = A{Int64}(undef, Cl), A{Int64}(undef, Cl), A{Int64}(undef, Cl), A{Int64}(undef, Cl), A(zeros(Int64, 1))
        pmap = cmap === nothing ? Ptr{Cvoid}(C_NULL) : Ptr{Cvoid}(pointer(cmap))
        check(ccall((:pdmp_sample_skeleton_until_device, LIB), Cint,
                    (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, UInt64, Cdouble, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                    s.handle, Cl, cap, chain_id_offset, pointer(xinit), pointer(vinit), seed_word(seed), T, skip, pmap, pointer(X), pointer(V), pointer(t), C_NULL, pointer(nrec), stream))
        check(ccall((:pdmp_close_chunk_device, LIB), Cint,
                    (Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                    s.handle, Cl, cap, pointer(X), pointer(V), pointer(t), C_NULL, T, pointer(nrec), pointer(count), stream))
        check(ccall((:pdmp_until_advance_device, LIB), Cint,
                    (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                    s.handle, Cl, pointer(count), skip, pmap, pointer(total), pointer(start), pointer(nxt), pointer(nact), stream))
        check_status(s)                                                          # waits for the chunk
        push!(chunks, (X=X, V=V, t=t, count=count, start=start, map=cmap, cap=cap, skip=skip, C=Cl))
        n = Array(nact)[1]
        n == 0 && break
        cmap, Cl, skip = nxt[1:n], n, 1                                          # the chains still running, in launch order
        cap_sum += cap
        # later chunks: the reference doubles its one chain's buffer (sample.jl:375-377); here the capacity grows with what has been used
        # so far -- at most four times the columns allocated so far, as in the Python mirror (which also sizes it from the event rates)
        cap = min(8 * cld(max(16, min(2 * cap, 4 * cap_sum)), 8), max(cap_limit, 8))
    end
    offsets = A{Int64}(undef, C + 1)
    check(ccall((:pdmp_compact_offsets_device, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), s.handle, C, pointer(total), pointer(offsets), stream))
    tot = Array(offsets)[end]
    Xo, Vo, to = A{Float64}(undef, d, tot), A{Float64}(undef, d, tot), A{Float64}(undef, tot)
    for ch in chunks
        pmap = ch.map === nothing ? Ptr{Cvoid}(C_NULL) : Ptr{Cvoid}(pointer(ch.map))
        check(ccall((:pdmp_compact_chunk_device, LIB), Cint,
                    (Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                    s.handle, ch.C, ch.cap, pointer(ch.count), ch.skip, pmap, pointer(ch.start), pointer(offsets), pointer(ch.X), pointer(ch.V), pointer(ch.t), C_NULL,
                    pointer(Xo), pointer(Vo), pointer(to), C_NULL, stream))
    end
    check_status(s)
    return (X=Xo, V=Vo, t=to, offsets=offsets, count=total)
end

# ---------------------------------------------------------------------------------------------- tape replay (julia/dump_tape.jl)
# Replace the engine's Philox stream by pre-recorded variates (one tape per kind, in consumption order): a dump of MersenneTwister's
# randexp / rand / randn calls made by the reference replays draw for draw, on whichever kernel form the launch uses otherwise
# (group kernels, one lane per chain; every sampler, ForwardECMC included).
function set_tape!(s::HIPSampler, exp_tape::Matrix{Float64}, uni_tape::Matrix{Float64}, nrm_tape::Matrix{Float64})   # columns = chains
    C = size(exp_tape, 2)
    GC.@preserve exp_tape uni_tape nrm_tape check(ccall((:pdmp_set_tape, LIB), Cint,
        (Ptr{Cvoid}, Int64, Int64, Int64, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
        s.handle, C, size(exp_tape, 1), size(uni_tape, 1), size(nrm_tape, 1), exp_tape, uni_tape, nrm_tape))
end
clear_tape!(s::HIPSampler) = check(ccall((:pdmp_clear_tape, LIB), Cint, (Ptr{Cvoid},), s.handle))

end # module
