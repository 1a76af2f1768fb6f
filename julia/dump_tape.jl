# dump_tape.jl -- produces REFERENCE-HELD fixtures for the parity tests (SURVEY.md 8(f2), tests/test_julia_fixtures.py).
#
# Run once, by anyone with Julia and PDMPFlux.jl v0.4.6 (neither exists in this pipeline: this file has never been executed here):
#
#     julia --project=/path/to/PDMPFlux.jl julia/dump_tape.jl tests/golden/julia
#
# For every case below it runs the reference's own sample_skeleton loop (src/sample.jl:253-284: init_state + get_event_state +
# record!) with the sampler's RNG replaced by a TapeRNG -- a thin AbstractRNG that forwards every call to the MersenneTwister(seed)
# the reference would have used (AbstractPDMP.jl:100) and logs the variates in call order, one tape per kind:
#     randexp(rng)                 SamplingLoopInplace.jl:70,136,190        -> exp.f64
#     rand(rng) / Categorical      SamplingLoopInplace.jl:154, ZigZagSamplers.jl:104, BouncyParticleSamplers.jl:54, BoomerangSamplers.jl:55 -> uni.f64
#     randn!(rng, v)               BouncyParticleSamplers.jl:67             -> nrm.f64
#     rand(key), randn(key, 2, dim), randn(key, dim)   ForwardEventChainMonteCarlo.jl:62,79,107,138,160,164 -> uni.f64 / nrm.f64
#                                  (randn(key, dims...) fills an array through randn!: logged column-major, i.e. g[1,j], g[2,j] adjacent)
# and writes the tapes and the resulting PDMPHistory as raw little-endian arrays plus a meta.json.  The engine (pdmp_set_tape) and
# the checker (orc_config_t.tape_*) replay the tapes draw for draw; tests/test_julia_fixtures.py compares their skeletons with the
# dumped history at the north-star tolerance.  (Boomerang's refresh draws `randn(dim)` from the GLOBAL rng, BoomerangSamplers.jl:65,
# SURVEY quirk Q8: not capturable through sampler.rng, so the Boomerang case runs with refresh_rate = 0.)
using Random, JSON
using PDMPFlux
import PDMPFlux: init_state, get_event_state, record!, PDMPHistory

mutable struct TapeRNG <: Random.AbstractRNG
    inner::Random.MersenneTwister
    exp::Vector{Float64}
    uni::Vector{Float64}
    nrm::Vector{Float64}
end
TapeRNG(seed::Int) = TapeRNG(Random.MersenneTwister(seed), Float64[], Float64[], Float64[])
# the three entry points the sampling loop uses; each forwards to the inner generator's own method and logs the result
Random.randexp(r::TapeRNG) = (x = randexp(r.inner); push!(r.exp, x); x)
Random.randexp(r::TapeRNG, ::Type{Float64}) = Random.randexp(r)
Random.rand(r::TapeRNG, ::Random.SamplerTrivial{Random.CloseOpen01{Float64}}) = (x = rand(r.inner); push!(r.uni, x); x)
Random.randn(r::TapeRNG, ::Type{Float64}=Float64) = (x = randn(r.inner); push!(r.nrm, x); x)
function Random.randn!(r::TapeRNG, A::AbstractArray{Float64})
    randn!(r.inner, A)                       # exactly what the reference executes with a plain MersenneTwister
    append!(r.nrm, vec(A))
    return A
end
# anything else would silently bypass the tapes: fail loudly instead
Random.rand(r::TapeRNG, sp::Random.Sampler) = error("TapeRNG: unlogged draw $(typeof(sp)) -- extend dump_tape.jl")

U_gauss(x) = sum(x .^ 2) / 2                                                   # README.md:36-38
function U_banana(x)                                                           # test/test_config.jl:33-36
    mean_x2 = x[1]^2 - 1
    return (x[1]^2 + (x[2] - mean_x2)^2 + sum(x[3:end] .^ 2)) / 2
end
grad_banana(x) = vcat(x[1] - 2x[1] * (x[2] - (x[1]^2 - 1)), x[2] - (x[1]^2 - 1), x[3:end])

# name, constructor, potential, dim, keywords, target descriptor of the engine, target parameters
diag_params(d) = (rs = MersenneTwister(7); (randn(rs, d), exp.(2 .* rand(rs, d) .- 1)))
function cases()
    mu, prec = diag_params(50)
    U_diag(x) = sum(prec .* (x .- mu) .^ 2) / 2
    mu6, prec6 = diag_params(6)
    grad_diag6(x) = prec6 .* (x .- mu6)
    return [
        ("zigzagad_iso_d6", ZigZagAD, U_gauss, 6, (;), "iso", Float64[]),
        ("zigzagad_banana_d6", ZigZagAD, U_banana, 6, (;), "banana", Float64[]),
        ("zigzagad_iso_d50", ZigZagAD, U_gauss, 50, (;), "iso", Float64[]),
        ("zigzagad_banana_d50", ZigZagAD, U_banana, 50, (;), "banana", Float64[]),
        ("zigzagad_diag_d50", ZigZagAD, U_diag, 50, (;), "diag", vcat(mu, prec)),
        ("zigzag_fd_iso_d6", ZigZag, x -> x, 6, (;), "iso", Float64[]),
        ("bps_banana_d6", BPS, grad_banana, 6, (;), "banana", Float64[]),
        ("bps_banana_d50", BPS, grad_banana, 50, (;), "banana", Float64[]),
        ("bps_errorpath_d2", BPS, grad_banana, 2, (; tmax=20.0, adaptive=false), "banana", Float64[]),
        ("boomerang_diag_d6_norefresh", Boomerang, grad_diag6, 6, (; refresh_rate=0.0), "diag", vcat(mu6, prec6)),
        # ForwardECMC: the reference's most-tested grid-path sampler (test/test_integration.jl:41-65, test/test_diagnostics.jl:128)
        ("forwardecmc_iso_d6", ForwardECMC, x -> x, 6, (;), "iso", Float64[]),
        ("forwardecmcad_banana_d6", ForwardECMCAD, U_banana, 6, (;), "banana", Float64[]),
        ("forwardecmc_iso_d50_ranp", ForwardECMC, x -> x, 50, (; ran_p=true), "iso", Float64[]),
        ("forwardecmc_diag_d6_fullrefresh", ForwardECMC, grad_diag6, 6, (; switch=false), "diag", vcat(mu6, prec6)),
    ]
end

function dump_case(outdir, name, ctor, pot, d, kw, target, tparams; n_sk=400, seed=20240501)
    sampler = ctor(d, pot; kw...)
    rs = MersenneTwister(sum(Int, collect(name)))
    xinit = target == "banana" ? 1 .+ 0.1 .* randn(rs, d) : randn(rs, d)
    vinit = ctor in (ZigZag, ZigZagAD) ? rand(rs, [-1.0, 1.0], d) : (v = randn(rs, d); ctor in (BPS, ForwardECMC, ForwardECMCAD) ? v ./ sqrt(sum(v .^ 2)) : v)
    x0, v0 = copy(xinit), copy(vinit)                      # init_state aliases its arguments (AbstractPDMP.jl:141-149)
    state = init_state(sampler, xinit, vinit, seed)
    tape = TapeRNG(seed)
    sampler.rng = tape                                     # same generator, same seed: only logged
    history = PDMPHistory(d, n_sk)
    record!(history, 1, state, d)
    for k in 2:n_sk
        state = get_event_state(state, sampler)
        record!(history, k, state, d)
    end
    dir = joinpath(outdir, name)
    mkpath(dir)
    put(file, a) = open(io -> write(io, htol.(a)), joinpath(dir, file), "w")
    put("xinit.f64", x0); put("vinit.f64", v0)
    put("exp.f64", tape.exp); put("uni.f64", tape.uni); put("nrm.f64", tape.nrm)
    put("X.f64", vec(history.X)); put("V.f64", vec(history.V)); put("t.f64", history.t)
    put("horizon.f64", history.horizon); put("ar.f64", history.ar)
    put("rejected.i32", history.rejected); put("hitting_horizon.i32", history.hitting_horizon); put("errored_bound.i32", history.errored_bound)
    meta = Dict("name" => name, "ctor" => string(ctor), "kwargs" => Dict(string(k) => v for (k, v) in pairs(kw)), "dim" => d,
                "target" => target, "target_params" => tparams, "seed" => seed, "n_sk" => n_sk,
                "n_exp" => length(tape.exp), "n_uni" => length(tape.uni), "n_nrm" => length(tape.nrm),
                "reference" => "PDMPFlux.jl " * string(pkgversion(PDMPFlux)), "julia" => string(VERSION))
    open(io -> JSON.print(io, meta, 1), joinpath(dir, "meta.json"), "w")
    println(name, ": ", n_sk, " events, tapes ", length(tape.exp), " / ", length(tape.uni), " / ", length(tape.nrm))
end

outdir = length(ARGS) >= 1 ? ARGS[1] : joinpath(@__DIR__, "..", "tests", "golden", "julia")
for c in cases()
    dump_case(outdir, c...)
end
