"""Multi-GPU path on CPU: world_size 2 over gloo.  The sharding / gather logic of
pdmpflux.jl_amd.ensemble.sample_skeleton_sharded is the product code; the per-shard compute is injected
(tests only) with the CPU oracle because this container has no GPU.  The contract checked: chains are
partitioned into contiguous ranges, keyed by GLOBAL chain id, and every gather mode reassembles exactly the
single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D, C_TOTAL, N_SK, N_SMP, SEED = 6, 7, 30, 11, 99          # 7 chains over 2 ranks: ragged shards (4 + 3)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _inputs():
    rs = np.random.default_rng(4)
    return rs.normal(size=(C_TOTAL, D)), rs.choice([-1.0, 1.0], size=(C_TOTAL, D))


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pdmpflux.jl_amd as P
    from pdmpflux.jl_amd import ensemble
    from oracle import oracle as O
    cfg = O.Config(sampler=O.ZIGZAG, dim=D)
    sampler = P.ZigZagAD(D, P.GaussianIso())

    def compute(xs, vs, chain0):            # stand-in for the HIP engine (tests only)
        r = O.sample_skeleton(cfg, N_SK, xs.numpy(), vs.numpy(), SEED, chain_id_offset=chain0, diagnostics=False)
        smp = O.sample_from_skeleton(cfg, N_SMP, r["X"], r["V"], r["t"])
        out = {k: torch.from_numpy(v) for k, v in r.items()}
        out["samples"] = torch.from_numpy(smp)
        return out

    x, v = _inputs()
    x, v = torch.from_numpy(x), torch.from_numpy(v)
    res = {}
    for mode in ("none", "skeleton", "samples", "moments"):
        r = ensemble.sample_skeleton_sharded(sampler, N_SK, x, v, seed=SEED, gather=mode, N_samples=N_SMP, compute=compute)
        res[mode] = {k: (val.numpy() if torch.is_tensor(val) else val) for k, val in r.items()}
    # the path bench.py --gpus N runs every step: preallocated full-size buffers, shard sampled into its rows, gathered in place;
    # ragged shards (7 chains: one broadcast per shard) and equal shards (the first 6 chains: one all-gather per array)
    for tag, n in (("inplace_ragged", C_TOTAL), ("inplace_equal", 6)):
        full = {"X": torch.zeros((n, N_SK, D), dtype=torch.float64), "V": torch.zeros((n, N_SK, D), dtype=torch.float64),
                "t": torch.zeros((n, N_SK), dtype=torch.float64)}
        r = ensemble.sample_skeleton_sharded(sampler, N_SK, x[:n], v[:n], seed=SEED, gather="skeleton", compute=compute, out_full=full)
        assert r["X"] is full["X"] and r["chain_range"] == (0, n)
        res[tag] = {k: full[k].numpy().copy() for k in ("X", "V", "t")}
    # the pipelined exchange (round 5, bench.py --pipeline): batches of chains, each batch sharded over the ranks, its all-gather issued
    # while the next batch is sampled; ragged batches and ragged shards inside a batch (7 chains in 2 and in 3 batches over 2 ranks)
    for tag, n, B in (("pipe2", C_TOTAL, 2), ("pipe3", C_TOTAL, 3), ("pipe3_equal", 6, 3), ("pipe_more_batches_than_chains", 3, 8)):
        full = {"X": torch.zeros((n, N_SK, D), dtype=torch.float64), "V": torch.zeros((n, N_SK, D), dtype=torch.float64),
                "t": torch.zeros((n, N_SK), dtype=torch.float64)}
        r = ensemble.sample_skeleton_sharded(sampler, N_SK, x[:n], v[:n], seed=SEED, gather="skeleton", compute=compute, out_full=full, pipeline=B)
        assert r["X"] is full["X"] and r["chain_range"] == (0, n) and 1 <= r["pipeline"] <= B
        res[tag] = {k: full[k].numpy().copy() for k in ("X", "V", "t")}
    np.save(os.path.join(outdir, f"rank{rank}.npy"), np.array([res], dtype=object), allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_batch_shard_ranges_partition_the_ensemble():
    from pdmpflux.jl_amd.ensemble import batch_shard_range
    for C_total, B, world in ((100000, 4, 8), (7, 3, 2), (6, 3, 2), (10, 4, 3), (5, 1, 2)):
        seen = []
        for j in range(B):
            for r in range(world):
                lo, hi = batch_shard_range(C_total, j, B, r, world)
                seen += list(range(lo, hi))
        assert seen == list(range(C_total))                                  # every chain exactly once, batch-major then rank-major


def test_sharded_ensemble_world2(tmp_path, oracle):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    x, v = _inputs()
    cfg = oracle.Config(sampler=oracle.ZIGZAG, dim=D)
    full = oracle.sample_skeleton(cfg, N_SK, x, v, SEED, diagnostics=False)
    full_s = oracle.sample_from_skeleton(cfg, N_SMP, full["X"], full["V"], full["t"])
    from pdmpflux.jl_amd.ensemble import shard_range
    for rank in range(world):
        res = np.load(os.path.join(str(tmp_path), f"rank{rank}.npy"), allow_pickle=True)[0]
        lo, hi = shard_range(C_TOTAL, rank, world)
        assert res["none"]["chain_range"] == (lo, hi)
        for k in ("X", "V", "t"):
            assert np.array_equal(res["none"][k], full[k][lo:hi]), (rank, k)           # shard = slice of the global run
            assert np.array_equal(res["skeleton"][k], full[k]), (rank, k)              # all-gather: everyone has everything
            assert np.array_equal(res["inplace_ragged"][k], full[k]), (rank, k)        # bench.py's in-place exchange, ragged shards
            assert np.array_equal(res["inplace_equal"][k], full[k][:6]), (rank, k)     # ... and equal shards
            assert np.array_equal(res["pipe2"][k], full[k]) and np.array_equal(res["pipe3"][k], full[k]), (rank, k)   # the pipelined exchange: the same bits
            assert np.array_equal(res["pipe3_equal"][k], full[k][:6]) and np.array_equal(res["pipe_more_batches_than_chains"][k], full[k][:3]), (rank, k)
        assert np.array_equal(res["samples"]["samples"], full_s)
        flat = full_s.reshape(-1, D)
        np.testing.assert_allclose(res["moments"]["mean"], flat.mean(0), rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(res["moments"]["var"], flat.var(0), rtol=1e-10)
        assert res["moments"]["count"] == flat.shape[0]
