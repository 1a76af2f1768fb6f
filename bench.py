#!/usr/bin/env python3
"""bench.py -- skeleton-events/sec of the sample_skeleton() hot path on MI355X.

Workload at N=1 (BASELINE.json configs[1]): Zig-Zag, d=50 standard Gaussian, 10^5 chains per GPU,
n_sk=1000 skeleton points per chain (9.99e7 events per step), synthetic stationary starts
(x ~ N(0,I), v = +-1), defaults of ZigZagAD (grid_size=10, tmax=2, signed vectorised bound,
adaptive, forward-mode slopes).  One "step" = one sample_skeleton() pass over the whole ensemble,
inputs and outputs resident in HBM.  With N>1 (torchrun, one rank per GPU) every rank runs its own
10^5 chains (weak scaling; chain ids are global) and each step ends with one RCCL all-gather of the
chains' end states.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (HBM, algorithmic
bytes 8(2d+1) per event over hipEvent-timed kernel durations) and `cpu_baseline` (the CPU oracle on
the host cores of the same box, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--chains", type=int, default=100_000, help="chains per GPU")
    ap.add_argument("--n-sk", type=int, default=1000)
    ap.add_argument("--dim", type=int, default=50)
    ap.add_argument("--sampler", default="ZigZagAD", choices=["ZigZagAD", "ZigZag", "BPS", "BPSAD", "Boomerang", "BoomerangAD", "ForwardECMC", "ForwardECMCAD", "SpeedUpZigZag", "SpeedUpZigZagAD", "RHMC"])
    ap.add_argument("--target", default="iso", choices=["iso", "diag", "banana", "dense", "src_banana"],
                    help="src_banana: U_banana given as source, differentiated by forward mode on the GPU (pdmp_create_from_source)")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline time; 0 disables it")
    return ap.parse_args()


def dense_precision(a):
    """BASELINE config 4: P = Q diag(lambda) Q', lambda log-uniform in [0.1, 10], Q from a QR (seed 7)."""
    import numpy as np
    rs = np.random.default_rng(7)
    Q, _ = np.linalg.qr(rs.normal(size=(a.dim, a.dim)))
    Pm = (Q * np.exp(rs.uniform(np.log(0.1), np.log(10.0), size=a.dim))) @ Q.T
    return 0.5 * (Pm + Pm.T)


def make_sampler(P, a):
    if a.target == "diag":                       # N(mu, diag(1/prec)), prec log-uniform in [1/e, e] (seed 7): a Boomerang that actually bounces
        import numpy as np
        rs = np.random.default_rng(7)
        return getattr(P, a.sampler)(a.dim, P.GaussianDiag(rs.normal(size=a.dim), np.exp(rs.uniform(-1, 1, size=a.dim))))
    tgt = {"iso": P.GaussianIso, "banana": P.Banana, "src_banana": lambda: P.SourcePotential(P.U_BANANA_SOURCE)}.get(a.target)
    tgt = tgt() if tgt else P.GaussianPrecision(dense_precision(a))
    return getattr(P, a.sampler)(a.dim, tgt)


def make_inputs(torch, a, dev, rank):
    g = torch.Generator(device=dev)
    g.manual_seed(a.seed * 1000 + rank)
    x = torch.randn((a.chains, a.dim), dtype=torch.float64, device=dev, generator=g)
    if a.target in ("banana", "src_banana"):
        x = 1.0 + 0.1 * x
    if a.sampler.startswith(("ZigZag", "SpeedUp")):
        v = torch.randint(0, 2, (a.chains, a.dim), device=dev, generator=g).to(torch.float64) * 2 - 1
    else:
        v = torch.randn((a.chains, a.dim), dtype=torch.float64, device=dev, generator=g)
        if a.sampler.startswith(("BPS", "ForwardECMC")):
            v = v / v.norm(dim=1, keepdim=True)
    return x.contiguous(), v.contiguous()


def cpu_baseline(a, sampler):
    """Oracle ('port' of the reference algorithm, reference summation order) on the host cores: a bounded
    sample of the same workload -- same d, n_sk, sampler and target, fewer chains."""
    import numpy as np
    from oracle import oracle as O
    cores = min(O.num_threads(), int(os.environ.get("PDMP_CPU_THREADS", "16")))   # CPU share of one GPU slot
    tid = {"iso": O.T_GAUSS_ISO, "diag": O.T_GAUSS_DIAG, "banana": O.T_BANANA, "dense": O.T_GAUSS_DENSE, "src_banana": O.T_AD_BANANA}[a.target]
    tparams = np.asfortranarray(dense_precision(a)).ravel(order="F") if a.target == "dense" else None
    if a.target == "diag":
        rs = np.random.default_rng(7)
        tparams = np.concatenate([rs.normal(size=a.dim), np.exp(rs.uniform(-1, 1, size=a.dim))])
    cfg = O.Config(target_params=tparams, sampler=sampler.sampler_id, dim=a.dim, grid_size=sampler.grid_size, tmax=sampler.tmax,
                   refresh_rate=sampler.refresh_rate, vectorized_bound=sampler.vectorized_bound,
                   signed_bound=sampler.signed_bound, adaptive=sampler.adaptive, deriv_mode=sampler.deriv_mode,
                   target_id=tid, arith=O.ARITH_REFERENCE, mix_p=getattr(sampler, "mix_p", 0.5), ran_p=getattr(sampler, "ran_p", False),
                   switch=getattr(sampler, "switch", True), positive=getattr(sampler, "positive", True),
                   step_size=getattr(sampler, "step_size", 0.05), phi=getattr(sampler, "phi", 1.5707963267948966))
    rs = np.random.default_rng(a.seed)
    n_sk = min(a.n_sk, 1000)

    def run(C, off):
        x = rs.normal(size=(C, a.dim))
        if a.target in ("banana", "src_banana"):
            x = 1.0 + 0.1 * x
        if a.sampler.startswith(("ZigZag", "SpeedUp")):
            v = rs.choice([-1.0, 1.0], size=(C, a.dim))
        else:
            v = rs.normal(size=(C, a.dim))
            v /= np.linalg.norm(v, axis=1, keepdims=True)
        t0 = time.perf_counter()
        O.sample_skeleton(cfg, n_sk, x, v, a.seed, chain_id_offset=off, diagnostics=False, n_threads=cores)
        return time.perf_counter() - t0

    C0 = 4 * cores
    dt = run(C0, 0)                                            # calibration (also warms the threads)
    per_chain = dt / C0
    cap = max(C0, int(1.5e9 // (n_sk * 8 * (2 * a.dim + 1))))  # keep the host buffers under ~1.5 GB
    total = max(C0, int(a.cpu_seconds / per_chain))
    done, spent, off = 0, 0.0, C0
    while done < total and spent < 2 * a.cpu_seconds:
        c = min(cap, total - done)
        spent += run(c, off)
        done += c
        off += c
    events = done * (n_sk - 1)
    return {"value": events / spent, "unit": "events/s", "cores": cores, "kind": "port",
            "sample": f"{done} chains x n_sk={n_sk} of the same workload ({events} events, {spent:.1f} s, "
                      f"OpenMP over chains, -O2 no fast-math, reference summation order)"}


def measured_traffic(a):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes (WRITE_SIZE exact, FETCH_SIZE x2
    on gfx950; collected by profiles/collect.sh in separate --pmc runs and committed as profiles/traffic.json).
    PMC counters cannot be read from inside this process, so the figure is the committed measurement of the
    identical workload, or null when none exists for it."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except (OSError, ValueError):
        return None
    key = f"{a.sampler}|{a.target}|d={a.dim}|C={a.chains}|n_sk={a.n_sk}"
    return table.get(key, {}).get("hbm_bytes_per_launch")


def main():
    a = parse()
    import torch
    import pdmpflux.jl_amd as P
    from pdmpflux.jl_amd import ensemble

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the engine has no CPU fallback)"
    # Rehearsal hook for a one-GPU box (never used by the driver): all ranks share device 0 and the exchange goes
    # through gloo on host copies, because RCCL refuses two ranks on one device.
    rehearsal = os.environ.get("PDMP_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    sampler = make_sampler(P, a)
    sampler.device = local
    x, v = make_inputs(torch, a, dev, rank)
    C, d, n_sk = a.chains, a.dim, a.n_sk
    out = {"X": torch.empty((C, n_sk, d), dtype=torch.float64, device=dev),
           "V": torch.empty((C, n_sk, d), dtype=torch.float64, device=dev),
           "t": torch.empty((C, n_sk), dtype=torch.float64, device=dev)}
    # the exchange is double-buffered: the all-gather of step i runs on RCCL's stream while the kernel of step i+1 runs
    # (the kernel's work units never wait on a wave that is not already running, so sharing CUs with RCCL cannot stall it)
    end_state = [torch.empty((C, 2 * d + 1), dtype=torch.float64, device=dev) for _ in range(2)] if world > 1 else None
    gathered = [torch.empty((world * C, 2 * d + 1), dtype=torch.float64, device=dev) for _ in range(2)] if world > 1 else None
    pending = [None, None]
    chain0 = rank * C

    def step(i):
        ensemble.sample_skeleton_device(sampler, n_sk, x, v, seed=a.seed + i, chain_id_offset=chain0, out=out,
                                        check=False)
        if world > 1:       # the path's only exchange: every rank learns every chain's end state
            b = i & 1
            if pending[b] is not None:          # the gather that last used this buffer pair (two steps ago)
                pending[b].wait()
                pending[b] = None
            es = end_state[b]
            es[:, :d] = out["X"][:, -1]
            es[:, d:2 * d] = out["V"][:, -1]
            es[:, 2 * d] = out["t"][:, -1]
            if rehearsal:
                parts = [torch.empty((C, 2 * d + 1), dtype=torch.float64) for _ in range(world)]
                dist.all_gather(parts, es.cpu())
                gathered[b].copy_(torch.cat(parts, 0))
            else:
                pending[b] = dist.all_gather_into_tensor(gathered[b], es, async_op=True)

    def fence():
        if world > 1:
            for b in (0, 1):                    # every gather issued so far is inside the timed region
                if pending[b] is not None:
                    pending[b].wait()
                    pending[b] = None
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    fence()
    ensemble.kernel_timing(sampler, reset=True)
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    from pdmpflux.jl_amd import _lib
    _lib.check(_lib.lib().pdmp_check_status(sampler.handle(), None))
    tm = ensemble.kernel_timing(sampler)
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    if world > 1:           # the last gather delivered this rank's own end states to its slot
        b = (a.warmup + a.steps - 1) & 1
        assert torch.equal(gathered[b][rank * C:(rank + 1) * C], end_state[b]), "all-gather of end states is inconsistent"
    # sanity inside the bench: strictly increasing event times on a slice, finite outputs
    tt = out["t"][:256]
    assert bool((tt[:, 1:] > tt[:, :-1]).all()) and bool(torch.isfinite(out["X"][:256]).all())

    if rank == 0:
        events_step = world * C * (n_sk - 1)
        value = events_step * a.steps / elapsed
        bytes_per_event = 8 * (2 * d + 1)                         # X, V, t of one event
        kernel_ms = tm["kernel_ms_total"] / max(tm["launches"], 1)
        achieved = C * (n_sk - 1) * bytes_per_event / (kernel_ms * 1e-3) / 1e9
        G = sampler.group_lanes()
        K = (d + G - 1) // G if G > 0 else 0
        # mirrors RegForm in csrc/group_kernel.hpp (which register-form instantiations exist)
        sid, fd = sampler.sampler_id, sampler.deriv_mode != 0
        reg_form = (G > 0 and sid in (0, 1, 2) and K <= 13 and G <= 16 and (G == 1 or K >= 7) and a.target in ("iso", "diag", "banana")
                    and not (sid == 0 and fd) and not (sid == 2 and K != 13) and os.environ.get("PDMP_REG", "1") != "0")
        kernel = (f"group_kernel_reg<{sampler.name},{a.target}> (G={G} lanes per chain, x and v in registers)" if reg_form else
                  f"group_kernel<{sampler.name},{a.target}> (G={G} lanes per chain)" if G > 0 else
                  f"skeleton_kernel<{sampler.name},{a.target}> (wave per chain)")
        res = {
            "metric": "skeleton-events/sec (whole node), Zig-Zag d=50, 10^5 chains",
            "value": value, "unit": "events/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{a.sampler} d={d} {a.target} target, {C} chains/GPU x n_sk={n_sk} "
                                   f"(BASELINE configs[1]), grid_size={sampler.grid_size}, tmax={sampler.tmax}, "
                                   f"adaptive, forward-mode slopes", "chains_per_gpu": C, "n_sk": n_sk, "dim": d,
                       "events_per_step": events_step,
                       "exchange": "none" if world == 1 else "all-gather of chain end states (2d+1 doubles per chain)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(a),
                         "kernel": kernel, "kernel_ms": kernel_ms, "bytes_per_event": bytes_per_event,
                         "note": "fp64 state machine bound by VALU issue (config 2: 115 VALU instructions per event, VALU busy 81 % "
                                 "at 2 waves/SIMD, profiles/r1/final3_zz_iso_pmc.json; Zig-Zag banana: 142, profiles/r1/zz_banana_r1b_pmc.json; "
                                 "BPS banana: 404, profiles/r1/bps_banana_r1b_pmc.json); HBM fraction is the requested headline"},
        }
        if a.target == "dense" and int(_lib.lib().pdmp_dense_batched_last_mode()) == 2:
            res["roofline"]["kernel"] = ("di_segment_kernel (wave per chain; P x and P v kept current by O(d) updates, one rocBLAS dgemm over all "
                                         "chains per 256 events re-bases them)")
            res["roofline"]["note"] = ("latency-bound: one chain per wave, 2 waves/SIMD; per event it writes 8(2d+1) B and reads one 8d-byte "
                                       "column of P (profiles/r1/c4_incr_pmc.json)")
        phases = int(_lib.lib().pdmp_dense_batched_last_phases()) if a.target == "dense" else 0
        if phases > 0:      # dense-precision path: the dominant kernel is the fp64 GEMM P x [requests | velocities] per phase
            cols = int(_lib.lib().pdmp_dense_batched_last_columns())   # C per phase, 2C when P v is re-synchronised (every 32nd phase)
            flops = 2.0 * d * d * cols
            tf = flops / (kernel_ms * 1e-3) / 1e12
            res["roofline"] = {"bound": "mfma", "achieved": tf, "peak": 78.6, "unit": "TFLOP/s", "frac": tf / 78.6, "traffic": None,
                               "kernel": "rocBLAS dgemm (v_mfma_f64) inside dense_batched", "kernel_ms": kernel_ms, "gemm_phases": phases,
                               "gemm_columns": cols,
                               "note": "achieved = 2 d^2 flop per multiplied column x columns / whole-launch time (GEMMs + step kernels); "
                                       "peak = MI355X fp64 matrix spec 78.6 TFLOP/s (datasheet; not in the local guide)"}
        if a.cpu_seconds > 0 and world == 1:                     # CPU baseline: rank 0 at N = 1 only
            res["cpu_baseline"] = cpu_baseline(a, sampler)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
