"""SpeedUpZigZag (src/Samplers/SpeedUpZigZagSamplers.jl: the Zig-Zag loop with the closed-form non-linear flow :71-79 and
grad-U_effective = speed grad-U - grad speed :81-83) on the group kernel against the checker, bit for bit."""
import numpy as np
import pytest

from helpers import assert_bitwise, make_init, make_pair

pytestmark = pytest.mark.gpu

CASES = [
    ("SpeedUpZigZagAD", 2, "iso", {}),
    ("SpeedUpZigZagAD", 5, "iso", {}),
    ("SpeedUpZigZagAD", 13, "diag", {}),
    ("SpeedUpZigZagAD", 50, "iso", {}),                              # G = 4
    ("SpeedUpZigZagAD", 20, "banana", {}),
    ("SpeedUpZigZag", 6, "iso", {}),                                 # default AD_backend = "FiniteDiff"
    ("SpeedUpZigZagAD", 30, "iso", {"signed_bound": False}),
    ("SpeedUpZigZagAD", 9, "diag", {"vectorized_bound": False}),
    ("SpeedUpZigZagAD", 100, "iso", {"tmax": 0.5, "adaptive": False}),
    ("SpeedUpZigZagAD", 7, "iso", {"grid_size": 0}),                 # Brent constant bound
]


@pytest.mark.parametrize("kind,dim,target,kw", CASES, ids=[f"{c[0]}-d{c[1]}-{c[2]}-{'-'.join(f'{k}={v}' for k, v in c[3].items())}" for c in CASES])
def test_speed_up_zigzag_matches_checker(engine, oracle, kind, dim, target, kw):
    s, cfg = make_pair(engine, oracle, kind, dim, target, **kw)
    C, n_sk = 17, 40
    x, v = make_init(kind, dim, C, seed=6, target=target)
    got = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=31, diagnostics=True)
    ref = oracle.sample_skeleton(cfg, n_sk, x, v, 31, diagnostics=True)
    assert_bitwise(got, ref)


def test_speed_up_zigzag_samples_use_the_nonlinear_flow(engine, oracle):
    """sample_from_skeleton interpolates with the sampler's own flow (sample.jl:499): bit for bit against the checker, and the
    samples of a long run have the moments of the target N(0, I) -- which checks flow and effective gradient together."""
    d, C = 4, 96
    s, cfg = make_pair(engine, oracle, "SpeedUpZigZagAD", d, "iso")
    x, v = make_init("SpeedUpZigZagAD", d, C, seed=2)
    sk = engine.sample_skeleton_ensemble(s, 3000, x, v, seed=12, diagnostics=False)
    smp = engine.sample_from_skeleton_ensemble(s, 3000, sk["X"], sk["V"], sk["t"])
    ref = oracle.sample_from_skeleton(cfg, 3000, sk["X"], sk["V"], sk["t"])
    assert np.array_equal(smp, ref)
    flat = smp.reshape(-1, d)
    assert np.abs(flat.mean(0)).max() < 0.05 and np.abs(flat.var(0) - 1).max() < 0.1


@pytest.mark.parametrize("kind,dim,target,kw", [("SpeedUpZigZagAD", 5, "iso", {}), ("RHMC", 7, "banana", {}), ("RHMC", 130, "diag", {"step_size": 0.1})])
def test_moments_and_time_horizon_use_the_samplers_own_flow(engine, kind, dim, target, kw):
    """pdmp_sample_moments_device = sequential sums over the samples of sample_from_skeleton, and pdmp_close_at_time_device = the
    closing point by flow from the last event, both with the non-linear / integrated flow of these samplers."""
    import torch
    from pdmpflux.jl_amd import ensemble
    ctor = getattr(engine, kind)
    rs = np.random.default_rng(dim)
    tg = {"iso": engine.GaussianIso(), "banana": engine.Banana(), "diag": engine.GaussianDiag(0.2 * rs.normal(size=dim), np.exp(rs.uniform(-1, 1, size=dim)))}[target]
    s = ctor(dim, tg, **kw)
    C, n_sk, N = 9, 60, 50
    x, v = make_init(kind, dim, C, seed=1, target=target)
    dev = torch.device("cuda:0")
    sk = ensemble.sample_skeleton_device(s, n_sk, torch.from_numpy(x).to(dev), torch.from_numpy(v).to(dev), seed=5)
    smp = ensemble.sample_from_skeleton_device(s, N, sk["X"], sk["V"], sk["t"]).cpu().numpy()
    s1, s2 = ensemble.sample_moments_device(s, N, sk["X"], sk["V"], sk["t"])
    r1, r2 = np.zeros((C, dim)), np.zeros((C, dim))
    for j in range(N):
        r1 = r1 + smp[:, j]
        r2 = r2 + smp[:, j] * smp[:, j]
    assert np.array_equal(s1.cpu().numpy(), r1) and np.array_equal(s2.cpu().numpy(), r2)
    # time-horizon cut: the closing column equals a sample_from_skeleton point at the same time
    T = float(sk["t"][:, -1].min()) * 0.61
    out = ensemble.sample_skeleton_until_device(s, T, torch.from_numpy(x).to(dev), torch.from_numpy(v).to(dev), seed=5, init_capacity=n_sk)
    cnt = out["count"].cpu().numpy()
    Xh, Vh, th = sk["X"].cpu().numpy(), sk["V"].cpu().numpy(), sk["t"].cpu().numpy()
    for c in range(C):
        k = int(cnt[c])
        assert out["t"][c, k - 1].item() == T
        one = engine.sample_from_skeleton_ensemble(s, 1, Xh[c:c + 1], Vh[c:c + 1], th[c:c + 1], discard_vt=False, dt=T)[0, 0]   # the point at 1 * dt = T
        assert np.array_equal(out["X"][c, k - 1].cpu().numpy(), one[:dim]) and np.array_equal(out["V"][c, k - 1].cpu().numpy(), one[dim:2 * dim])


@pytest.mark.parametrize("dim,target,seg", [(2, "iso", 0), (5, "iso", 3), (13, "diag", 0), (50, "iso", 4), (20, "banana", 0), (50, "banana", 2), (48, "diag", 0),
                                            (100, "iso", 0), (7, "banana", 0), (24, "diag", 5), (64, "banana", 0)])
def test_speed_up_zigzag_register_and_lds_forms_agree(engine, oracle, monkeypatch, dim, target, seg):
    """Round 4: SpeedUpZigZag in the register form (exact-K kernels for the slot counts that stay free of spills: K <= 7 on the isotropic
    Gaussian, 6 on the diagonal one, 4 on the banana -- inst_reg_suzz_k*.hip; forward-mode slopes, the default bound flags): the same
    bits as the LDS form and the checker, hand-off segments included."""
    if seg:
        monkeypatch.setenv("PDMP_FORCE_SEGMENTS", str(seg))
    outs = []
    for reg in ("1", "0"):
        monkeypatch.setenv("PDMP_REG", reg)
        s, cfg = make_pair(engine, oracle, "SpeedUpZigZagAD", dim, target)
        K = -(-dim // s.group_lanes())
        assert 2 <= K <= {"iso": 7, "diag": 6, "banana": 4}[target]
        x, v = make_init("SpeedUpZigZagAD", dim, 37, seed=5, target=target)
        outs.append(engine.sample_skeleton_ensemble(s, 60, x, v, seed=13, diagnostics=True))
        assert s.last_launch_form() == (3 if reg == "1" else 2)
    assert_bitwise(outs[0], outs[1])
    assert_bitwise(outs[0], oracle.sample_skeleton(cfg, 60, x, v, 13, diagnostics=True))
