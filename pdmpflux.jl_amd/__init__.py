"""pdmpflux.jl_amd -- MI355X-native engine for PDMPFlux.jl's sample_skeleton()/sample_from_skeleton()
hot path (ensembles of independent Zig-Zag / BPS / Boomerang chains), behind the reference's own
sampler API.  Compute happens only in libpdmphip.so (hand-written HIP for gfx950, C ABI in
include/pdmp_hip.h); there is no CPU fallback.
"""
from ._lib import ArgumentError, DimensionMismatch, EngineError, LIB_PATH
from .targets import (GaussianIso, GaussianDiag, Banana, GaussianPrecision, SourcePotential, SourceGradient,
                      U_BANANA_SOURCE, U_FUNNEL_SOURCE, U_RIDGED_GAUSS_SOURCE)
from .samplers import AbstractPDMP, ZigZag, ZigZagAD, BPS, BPSAD, Boomerang, BoomerangAD, ForwardECMC, ForwardECMCAD, SpeedUpZigZag, SpeedUpZigZagAD, RHMC, RHMCAD, StickyZigZag, StickyZigZagAD
from .history import PDMPHistory
from .sample import (sample, sample_skeleton, sample_from_skeleton, sample_skeleton_ensemble,
                     sample_from_skeleton_ensemble, sample_skeleton_until)

__all__ = [
    "ArgumentError", "DimensionMismatch", "EngineError", "LIB_PATH",
    "GaussianIso", "GaussianDiag", "Banana", "GaussianPrecision", "SourcePotential", "SourceGradient",
    "U_BANANA_SOURCE", "U_FUNNEL_SOURCE", "U_RIDGED_GAUSS_SOURCE",
    "AbstractPDMP", "ZigZag", "ZigZagAD", "BPS", "BPSAD", "Boomerang", "BoomerangAD", "ForwardECMC", "ForwardECMCAD", "SpeedUpZigZag", "SpeedUpZigZagAD", "RHMC", "RHMCAD", "StickyZigZag", "StickyZigZagAD",
    "PDMPHistory", "sample", "sample_skeleton", "sample_from_skeleton",
    "sample_skeleton_ensemble", "sample_from_skeleton_ensemble", "sample_skeleton_until",
]
