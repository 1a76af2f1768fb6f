"""pdmp_init_states_device -- the synthetic initial states of BASELINE.md section 3 ("xinit[:,c] ~ N(0,I), vinit[i,c] = +-1 from Philox
key (seed, c) stream 1; chain RNG = stream 0") generated on the device -- against the checker's orc_init_states, bit for bit; keyed by
the global chain id (a shard's states do not depend on the sharding); distinct from the chains' own stream 0."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dim", [1, 2, 7, 50, 51, 100])
@pytest.mark.parametrize("v_kind", [0, 1, 2])
def test_init_states_bitwise(engine, oracle, dim, v_kind):
    import torch
    from pdmpflux.jl_amd import ensemble
    s = engine.ZigZagAD(dim, engine.GaussianIso())
    C, seed = 193, 2024
    x, v = ensemble.init_states_device(s, C, seed=seed, x_mean=1.0, x_scale=0.1, v_kind=v_kind)
    torch.cuda.synchronize()
    rx, rv = oracle.init_states(seed, C, dim, x_mean=1.0, x_scale=0.1, v_kind=v_kind)
    assert np.array_equal(x.cpu().numpy(), rx) and np.array_equal(v.cpu().numpy(), rv)
    if v_kind == 0:
        assert set(np.unique(rv)) <= {-1.0, 1.0}
    if v_kind == 2 and dim > 1:
        assert np.allclose((rv * rv).sum(1), 1.0, atol=1e-14)


def test_init_states_sharding_invariance_and_stream_separation(engine, oracle):
    import torch
    from pdmpflux.jl_amd import ensemble
    d = 50
    s = engine.ZigZagAD(d, engine.GaussianIso())
    x, v = ensemble.init_states_device(s, 1000, seed=7)
    x2, v2 = ensemble.init_states_device(s, 300, seed=7, chain_id_offset=450)
    assert torch.equal(x[450:750], x2) and torch.equal(v[450:750], v2)
    # stream 1 is not stream 0: the first normals of the chain's own stream (what a BPS refresh would draw) differ
    own = oracle.rng_dump(7, 0, 2, d)
    assert not np.array_equal(own, x[0].cpu().numpy())
    # moments of the stationary start
    xs = ensemble.init_states_device(s, 20000, seed=11)[0].cpu().numpy()
    assert abs(xs.mean()) < 5e-3 and abs(xs.var() - 1.0) < 5e-3
    with pytest.raises(engine.ArgumentError):
        ensemble.init_states_device(s, 0, seed=1)
