"""bench.py's sharded path itself under test (the driver launches it on N GPUs at round end; a test box has one): two ranks started by
`python -m torch.distributed.run`, both on GPU 0, the exchange through gloo on host copies (PDMP_BENCH_REHEARSAL=1 -- RCCL refuses two ranks
on one device).  Same code as the N-GPU run: contiguous chain ranges, global chain ids, the in-place gather of X, V, t, max-over-ranks
timing, rank 0's JSON line; checked against a one-rank run of the same small workload."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env):
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                    # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_two_ranks_rehearsal():
    common = ["--steps", "2", "--warmup", "1", "--chains", "3001", "--n-sk", "64", "--cpu-seconds", "0", "--pmc", "off"]
    env = dict(os.environ, PDMP_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29731", "bench.py", "--gpus", "2"] + common, env)
    one = _run([sys.executable, "bench.py", "--gpus", "1"] + common, dict(os.environ))
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["scaling"] == "strong" and two["unit"] == "events/s" and two["metric"] == one["metric"]
    c2 = two["config"]
    assert c2["chains"] == 3001 and c2["chains_rank0"] == 1501                      # four batches of 751 / 750 chains, rank 0 takes 376 + 375 + 375 + 375
    assert c2["pipeline_batches"] == 4 and c2["kernel_launches_per_step"] == 4
    assert c2["events_per_step"] == 3001 * 63 == one["config"]["events_per_step"]
    assert c2["exchange_bytes_per_gpu"] == 1500 * 64 * 8 * (2 * 50 + 1)             # rank 0 receives the other shard's X, V, t
    assert two["value"] > 0 and two["roofline"]["kernel_ms"] > 0 and c2["exchange_ms"] > 0
    assert "hipMalloc" in c2["allocator"]                                           # the exchange buffers come from torch's allocator


@pytest.mark.parametrize("extra", [[], ["--pipeline", "1"], ["--gather", "moments", "--n-samples", "32"], ["--gather", "samples", "--n-samples", "16"],
                                   ["--gather", "none"]], ids=["pipelined", "unpipelined", "moments", "samples", "none"])
def test_bench_launches_its_own_ranks(extra):
    """Round 5: `python3 bench.py --gpus 2` WITHOUT a launcher around it (the way the driver calls `--gpus 1`) starts its two ranks itself,
    before touching the GPU, and prints rank 0's one JSON line; here as a rehearsal on one GPU (both ranks on device 0, gloo)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PDMP_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--chains", "2000", "--n-sk", "48", "--cpu-seconds", "0",
                "--pmc", "off"] + extra, env)
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["value"] > 0
    c = res["config"]
    gather = extra[1] if extra and extra[0] == "--gather" else "skeleton"
    assert c["gather"] == gather and c["kernel_only_events_per_s"] > 0
    if gather == "skeleton":
        assert c["pipeline_batches"] == (1 if "--pipeline" in extra else 4) and c["exchange_ms"] > 0
    if gather == "moments":
        assert c["exchange_bytes_per_gpu"] == 8 * 101
    if gather == "samples":
        assert c["exchange_bytes_per_gpu"] == 1000 * 16 * 50 * 8


def test_bench_ab_env_times_both_values_on_one_allocation():
    """`bench.py --ab-env NAME=v1,v2`: after the timed region the same arrays are launched on again under each value of a launch-policy
    knob (the handle re-reads its policy in between), so that two settings can be compared without the +-5 % that the placement of two
    processes' allocations puts between them.  Here the number of work units per resident wave of the lane kernel (same results)."""
    r = _run([sys.executable, "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--chains", "40000", "--n-sk", "64",
              "--cpu-seconds", "0", "--pmc", "off", "--placement-survey", "off", "--ab-env", "PDMP_LANE_ROUNDS=8,24"], dict(os.environ))
    ab = r["ab_env"]
    assert ab["knob"] == "PDMP_LANE_ROUNDS" and set(ab["ms"]) == {"8", "24"}
    assert all(len(v) == 6 and min(v) > 0 for v in ab["ms"].values())             # three rounds of two launches per value, interleaved
    assert set(ab["mean_ms"]) == {"8", "24"}
