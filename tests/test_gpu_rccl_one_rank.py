"""What a one-GPU box can exercise of the N > 1 exchange of `bench.py --gpus N` / `ensemble.sample_skeleton_sharded`: the torch.distributed
backend the multi-GPU run uses (`nccl` = RCCL) with ONE rank -- the in-place all-gather whose input is a slice of its own output, on
sub-ranges of the arrays (the batches of the pipelined exchange), asynchronously, with a kernel launched between the call and the wait.
Run in a child process: a process group is process-wide state the rest of the suite must not inherit."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
    import numpy as np, torch, torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29791", RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import pdmpflux.jl_amd as P
    from pdmpflux.jl_amd import ensemble
    from helpers import make_init
    d, C, n_sk = 50, 1003, 40
    x, v = make_init("ZigZagAD", d, C, seed=1)
    x, v = torch.from_numpy(x).to(dev), torch.from_numpy(v).to(dev)
    s = P.ZigZagAD(d, P.GaussianIso())
    ref = ensemble.sample_skeleton_device(s, n_sk, x, v, seed=5)
    torch.cuda.synchronize()
    # (1) the collective itself, the way the pipelined exchange calls it: in place on a sub-range of rows, asynchronous, work queued behind it
    full = {k: ref[k].clone() for k in ("X", "V", "t")}
    works = []
    for a, b in ((0, 251), (251, 502), (502, 753), (753, 1003)):
        for k in ("X", "V", "t"):
            works += ensemble._gather_in_place(full[k][a:b], b - a, 0, 1, None, async_op=True)
        junk = torch.empty(1 << 20, device=dev).normal_()            # a kernel between the call and the wait
    assert len(works) == 12
    for w in works:
        w.wait()
    torch.cuda.synchronize()
    for k in ("X", "V", "t"):
        assert torch.equal(full[k], ref[k]), k                        # one rank: the gather must leave the rows as they are
    # (2) the pipelined step of bench.py on this backend (one rank: four launches, no exchange) against the plain one: the same bits
    out = {"X": torch.zeros((C, n_sk, d), dtype=torch.float64, device=dev), "V": torch.zeros((C, n_sk, d), dtype=torch.float64, device=dev),
           "t": torch.zeros((C, n_sk), dtype=torch.float64, device=dev)}
    r = ensemble.sample_skeleton_sharded(s, n_sk, x, v, seed=5, gather="skeleton", out_full=out, pipeline=4)
    torch.cuda.synchronize()
    assert r["pipeline"] == 4
    for k in ("X", "V", "t"):
        assert torch.equal(out[k], ref[k]), k
    # (3) the moments exchange: one all-reduce of 2 d + 1 doubles on the device
    m = ensemble.sample_skeleton_sharded(s, n_sk, x, v, seed=5, gather="moments", N_samples=16)
    assert float(m["count"]) == C * 16 and bool(torch.isfinite(m["mean"]).all())
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_ONE_RANK_OK")
''')


def test_inplace_async_allgather_and_pipelined_step_on_rccl():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, "-c", CHILD], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "RCCL_ONE_RANK_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
