#!/usr/bin/env python3
"""bench.py -- skeleton-events/sec of the sample_skeleton() hot path on MI355X.

Default workload (BASELINE.json configs[1], the configuration the metric is quoted on): Zig-Zag, d=50 standard Gaussian,
ONE ensemble of 10^5 chains, n_sk=1000 skeleton points per chain (9.99e7 events per step), synthetic stationary starts
(x ~ N(0,I), v = +-1), defaults of ZigZagAD (grid_size=10, tmax=2, signed vectorised bound, adaptive, forward-mode slopes).
One "step" = one sample_skeleton() pass over the whole ensemble, inputs and outputs resident in HBM.

With N>1 (torchrun, one rank per GPU) the SAME ensemble is sharded over the ranks (strong scaling: rank r takes the contiguous
chain range shard_range(chains, r, N), chain ids stay global) and every step ends, inside the timed region, with the north
star's exchange: one RCCL all-gather each for the X, V and t shards, after which every GPU holds the whole skeleton
(`ensemble.sample_skeleton_sharded(gather="skeleton")`, in place in preallocated full-size buffers).  `--gather none` times
the sharded kernels alone.  The lanes-per-chain layout is ONE for all ranks (`--lanes auto` = pdmp_suggest_group_lanes, which returns
the default layout for every shard size: more lanes per chain were measured and do not shorten a small shard's launch).

`--driver multi` runs the same sharded step through the library's own multi-device entry (`pdmp_multi_sample_skeleton`,
csrc/multi_device.hip: ONE process, one stream per device, grouped in-place ncclAllGather) -- the entry a Julia host uses -- instead
of one process per GPU over torch.distributed: `python bench.py --gpus N --driver multi` (no torchrun).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (HBM, algorithmic
bytes 8(2d+1) per event over hipEvent-timed kernel durations; `roofline.valu`: the VALU-issue ceiling that actually binds)
and `cpu_baseline` (the CPU oracle on all host cores of the same box, bounded sample).
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
    ap.add_argument("--chains", type=int, default=100_000, help="chains of the ensemble (sharded over the GPUs)")
    ap.add_argument("--gather", default="skeleton", choices=["skeleton", "none", "moments", "samples"],
                    help="N > 1: the exchange that ends every step.  skeleton: all-gather of X, V, t -- the north star's, 808 B per event; "
                         "samples: sample_from_skeleton(--n-samples) on every shard, all-gather of the samples; moments: the fused moment sums of "
                         "those samples, one all-reduce of 2 dim + 1 doubles")
    ap.add_argument("--n-samples", type=int, default=0, help="--gather samples / moments: samples per chain (0: n_sk)")
    ap.add_argument("--pipeline", default="auto",
                    help="N > 1, --gather skeleton: batches of chains whose all-gather overlaps the sampling of the next batch "
                         "(ensemble.sample_skeleton_sharded(pipeline=B)); auto = 4, 1 = the exchange after the whole shard")
    ap.add_argument("--lanes", default="auto", help="lanes per chain: auto (pdmp_suggest_group_lanes: the default layout for every shard size), default, or a power of two")
    ap.add_argument("--driver", default="ranks", choices=["ranks", "multi"],
                    help="ranks: one process per GPU over torch.distributed (what the round driver launches); multi: ONE process, "
                         "pdmp_multi_sample_skeleton over --gpus devices (the library's own RCCL path)")
    ap.add_argument("--n-sk", type=int, default=1000)
    ap.add_argument("--dim", type=int, default=50)
    ap.add_argument("--sampler", default="ZigZagAD", choices=["ZigZagAD", "ZigZag", "BPS", "BPSAD", "Boomerang", "BoomerangAD", "ForwardECMC", "ForwardECMCAD", "SpeedUpZigZag", "SpeedUpZigZagAD", "RHMC", "StickyZigZag", "StickyZigZagAD"])
    ap.add_argument("--refresh-rate", type=float, default=None, help="BPS / Boomerang refresh_rate (default: the constructor's)")
    ap.add_argument("--target", default="iso", choices=["iso", "diag", "banana", "dense", "src_banana"],
                    help="src_banana: U_banana given as source, differentiated by forward mode on the GPU (pdmp_create_from_source)")
    ap.add_argument("--grid-size", type=int, default=None, help="grid_size of the bound (default: the constructor's 10; 0: Brent; any 2 .. 64 runs the LDS-form group kernel)")
    ap.add_argument("--rng-version", type=int, default=0, help="arithmetic contract of the random stream / elementary functions: 0 = default (2), 1 = rounds 1-4 (LDS-form kernels only)")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--ab-env", default=None, metavar="NAME=v1,v2[,..]",
                    help="after the timed region: hipEvent-timed launches on the SAME arrays under each value of a PDMP_* knob of the launch policy "
                         "(the handle re-reads it: pdmp_reload_policy), interleaved -- two processes differ by their allocations' placement (~5 %%)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline time; 0 disables it")
    ap.add_argument("--alloc", default="auto", choices=["auto", "chunked", "hipmalloc"],
                    help="allocator of the output arrays: chunked = ensemble.alloc_outputs (pdmp_device_alloc: independent 1-GiB physical chunks, "
                         "DESIGN.md 6), hipmalloc = torch.empty; auto: chunked at N = 1, hipmalloc for N > 1 (the exchange buffers go through RCCL)")
    ap.add_argument("--sets", type=int, default=1, help="allocation sets of X, V, t alive at once that the timed steps rotate over (three fit at the default size)")
    ap.add_argument("--placement-survey", default="auto", choices=["auto", "off"],
                    help="auto (N = 1, >= 5 steps): after the timed region and its checks, free the arrays and time five more FRESH allocations one "
                         "after the other -- three from the product's allocator, two from hipMalloc (reported in config.placement_ms, never in `value`)")
    ap.add_argument("--pmc", default="auto", choices=["auto", "off"],
                    help="auto (N = 1): measure roofline.traffic live -- child runs of this command (one launch each) under rocprofv3 --pmc "
                         "WRITE_SIZE, FETCH_SIZE and the VALU counters, before this process touches the GPU; off: quote profiles/traffic.json")
    return ap.parse_args()


def live_counters(a):
    """HBM bytes and VALU counters of ONE launch of this very workload on this GPU: three child processes `rocprofv3 --pmc ... --
    python3 bench.py --steps 1 --warmup 0 --pmc off <same workload>` (each counter group in its own run, kernel trace off, as
    MI355X_MICROARCH.md prescribes; WRITE_SIZE / FETCH_SIZE in KiB, FETCH_SIZE x2 on gfx950), run BEFORE this process initialises the
    GPU.  Returns None when rocprofv3 is missing or a pass fails: the line then quotes the committed measurement instead."""
    import csv, glob, shutil, subprocess, tempfile
    if shutil.which("rocprofv3") is None or os.environ.get("PDMP_BENCH_PMC") == "0":
        return None
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None                                              # this run is itself being profiled: no nested profiler
    args = ["--steps", "1", "--warmup", "0", "--cpu-seconds", "0", "--pmc", "off", "--chains", str(a.chains), "--n-sk", str(a.n_sk),
            "--dim", str(a.dim), "--sampler", a.sampler, "--target", a.target, "--seed", str(a.seed), "--lanes", a.lanes]
    if a.grid_size is not None:
        args += ["--grid-size", str(a.grid_size)]
    if a.refresh_rate is not None:
        args += ["--refresh-rate", str(a.refresh_rate)]
    out, env = {}, dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)                                         # the children are plain single-process runs
    for group in (["WRITE_SIZE"], ["FETCH_SIZE"], ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"]):
        tmp = tempfile.mkdtemp(prefix="pdmp_pmc_", dir="/tmp")
        try:
            # the interpreter itself after `--` (no env / launcher hop: the profiler has initialised the GPU before the program starts);
            # its own session, so that a timeout ends the profiled grandchild too instead of leaving it on the GPU
            cmd = ["rocprofv3", "--pmc", *group, "--output-format", "csv", "-d", tmp, "--", sys.executable, os.path.join(ROOT, "bench.py"), *args]
            proc = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = proc.wait(timeout=120)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
                proc.wait()
                return None
            files = glob.glob(os.path.join(tmp, "**", "*_counter_collection.csv"), recursive=True)
            if rc != 0 or not files:
                return None
            names = set()
            for row in csv.DictReader(open(files[0])):
                if "pdmp::" in row["Kernel_Name"]:
                    out[row["Counter_Name"]] = out.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                    names.add(row["Kernel_Name"].split("(")[0])
            out["kernels"] = sorted(names)
        except Exception:
            return None
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    if "WRITE_SIZE" not in out or "FETCH_SIZE" not in out:
        return None
    return out


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
    kw = {} if a.refresh_rate is None else {"refresh_rate": a.refresh_rate}
    if a.grid_size is not None:
        kw["grid_size"] = a.grid_size
    if a.rng_version:
        kw["rng_version"] = a.rng_version
    if a.sampler.startswith("Sticky"):           # spike-and-slab: thawing rate 0.5 for every coordinate (tests/test_gpu_sticky.py's regime)
        import numpy as np
        return getattr(P, a.sampler)(a.dim, tgt, np.full(a.dim, 0.5), **kw)
    return getattr(P, a.sampler)(a.dim, tgt, **kw)


def init_kind(a):
    """(x_mean, x_scale, v_kind) of the synthetic initial states, BASELINE.md section 3: stationary-ish starts x ~ N(0, I) (banana:
    ones + 0.1 N(0, I)); v: +-1 coordinates for the Zig-Zag family, uniform on the unit sphere for BPS / ForwardECMC, N(0, I) otherwise."""
    xm, xs = (1.0, 0.1) if a.target in ("banana", "src_banana") else (0.0, 1.0)
    vk = 0 if a.sampler.startswith(("ZigZag", "SpeedUp", "Sticky")) else (2 if a.sampler.startswith(("BPS", "ForwardECMC")) else 1)
    return xm, xs, vk


def make_inputs(ensemble, sampler, a, dev):
    """The WHOLE ensemble's initial states on every rank (16 d bytes per chain), generated on the device from stream 1 of each chain's
    Philox key (seed, c) -- BASELINE.md section 3; the chains' own draws come from stream 0 of the same key (pdmp_init_states_device)."""
    xm, xs, vk = init_kind(a)
    return ensemble.init_states_device(sampler, a.chains, seed=a.seed, x_mean=xm, x_scale=xs, v_kind=vk, device=dev)


def cpu_baseline(a, sampler):
    """Oracle ('port' of the reference algorithm, reference summation order) on ALL host cores of this box: a bounded
    sample of the same workload -- same d, n_sk, sampler and target, fewer chains.  Each OpenMP thread runs a contiguous slab
    of chains into its own part of the output (first touched by that thread), see oracle/pdmp_oracle.c."""
    import numpy as np
    from oracle import oracle as O
    nproc = os.cpu_count() or 1
    try:
        nproc = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    share = nproc                                   # the cores this process may actually use: the cgroup CPU quota, if there is one
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: [t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()])):
        try:
            q, per = parse(open(path).read())
            if q != "max" and int(q) > 0:
                share = max(1, min(nproc, -(-int(q) // int(per))))
            break
        except (OSError, ValueError):
            continue
    # measured on the GPU box (128 hardware threads visible, CPU share of one GPU slot = 16): 16 threads 3.4e6 events/s, 128 threads
    # 3.1e6 -- more threads than the share only add contention, so the baseline uses every core of the share
    cores = int(os.environ.get("PDMP_CPU_THREADS", "0")) or min(O.num_threads(), share)
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
    n_sk = min(a.n_sk, 1000)

    bufs = {}

    def run(C, off):
        if not bufs or bufs["t"].shape[0] < C:       # output buffers allocated and touched once, reused by every timed run
            bufs.update(X=np.zeros((C, n_sk, a.dim)), V=np.zeros((C, n_sk, a.dim)), t=np.zeros((C, n_sk)))
        out = {k: bufs[k][:C] for k in ("X", "V", "t")}
        xm, xs, vk = init_kind(a)                    # the same initial states the GPU run starts chains off .. off + C from (stream 1)
        x, v = O.init_states(a.seed, C, a.dim, chain_id_offset=off, x_mean=xm, x_scale=xs, v_kind=vk)
        t0 = time.perf_counter()
        O.sample_skeleton(cfg, n_sk, x, v, a.seed, chain_id_offset=off, diagnostics=False, n_threads=cores, out=out)
        return time.perf_counter() - t0

    # all cores of the share, and -- because a GPU slot's CPU share can be smaller than what the affinity mask shows -- 16 threads
    # as well: a short calibration decides, both rates are reported
    tried = {}
    for cand in sorted({cores, min(cores, 16)}):
        cores = cand
        run(4 * cand, 0)                                       # warms the threads
        tried[cand] = 4 * cand * (n_sk - 1) / run(4 * cand, 0)
    cores = max(tried, key=tried.get)
    C0 = 4 * cores
    cap = max(C0, int(3e9 // (n_sk * 8 * (2 * a.dim + 1))))    # keep the host buffers under ~3 GB
    cap -= cap % cores
    dt = run(C0, 0)                                            # calibration
    per_chain = dt / C0
    total = max(C0, int(a.cpu_seconds / per_chain))
    total = min(total, 40 * cap)
    done, spent, off = 0, 0.0, C0
    while done < total and spent < 2 * a.cpu_seconds:
        c = min(cap, total - done)
        c = max(cores, c - c % cores)                          # equal slabs
        spent += run(c, off)
        done += c
        off += c
    events = done * (n_sk - 1)
    return {"value": events / spent, "unit": "events/s", "cores": cores, "nproc": nproc, "cpu_share": share, "threads_tried": {str(k): v for k, v in tried.items()}, "kind": "port",
            "label": "CPU restatement of PDMPFlux.jl's algorithm (oracle/pdmp_oracle.c), not the Julia package itself",
            "sample": f"{done} chains x n_sk={n_sk} of the same workload ({events} events, {spent:.1f} s, "
                      f"OpenMP over chains on {cores} threads (hardware threads visible: {nproc}, CPU share of this process: {share}), static slabs, -O2 no fast-math, reference summation order)"}


def measured_counters(a, C_local, G, form=0):
    """HBM bytes per launch and VALU instruction counts of the dominant kernel from the rocprofv3 PMC passes (WRITE_SIZE exact,
    FETCH_SIZE x2 on gfx950; collected by profiles/collect.sh in separate --pmc runs and committed as profiles/traffic.json).
    PMC counters cannot be read from inside this process, so the figures are the COMMITTED measurement of the identical
    workload and layout (the entry names its source file), or None when none exists for it."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except (OSError, ValueError):
        return None
    key = f"{a.sampler}|{a.target}|d={a.dim}|C={C_local}|n_sk={a.n_sk}"
    if form in (4, 7):                                           # the lane kernel (one lane / a pair of lanes per chain) has its own measurements
        return table.get(key + "|lane")
    e = table.get(key + f"|G={G}") or table.get(key)
    return e


def metric_label(a, sampler, C_total):
    """BASELINE.json's metric wording, built from the arguments: the default run reads 'skeleton-events/sec (whole node), Zig-Zag d=50,
    10^5 chains'; any other sampler / target / slope mode / size says so."""
    family = "Zig-Zag" if a.sampler.startswith("ZigZag") else ("Boomerang" if a.sampler.startswith("Boomerang") else
             ("BPS" if a.sampler.startswith("BPS") else a.sampler))
    extra = "" if a.target == "iso" else f" {a.target}"
    if sampler.deriv_mode != 0:
        extra += " (finite-difference slopes)"
    chains = "10^5" if C_total == 100_000 else ("10^4" if C_total == 10_000 else str(C_total))
    nsk = "" if a.n_sk == 1000 else f", n_sk={a.n_sk}"
    if sampler.grid_size != 10 and not a.sampler.startswith("RHMC"):
        extra += f" (grid_size={sampler.grid_size})"
    return f"skeleton-events/sec (whole node), {family} d={a.dim}{extra}, {chains} chains{nsk}"


def workload_label(a, sampler, world, C_local):
    rr = f", refresh_rate={sampler.refresh_rate}" if a.sampler.startswith(("BPS", "Boomerang", "RHMC")) else ""
    slopes = "forward-mode slopes" if sampler.deriv_mode == 0 else "finite-difference slopes"
    base = {("ZigZagAD", "iso", 50, 100_000, 1000): " = BASELINE configs[1]",
            ("BPS", "banana", 50, 100_000, 1000): " = BASELINE configs[2]",
            ("BPSAD", "banana", 50, 100_000, 1000): " = BASELINE configs[2]",
            ("ZigZagAD", "dense", 1000, 10_000, 200): " = BASELINE configs[3] at n_sk=200",
            ("Boomerang", "iso", 100, 100_000, 1000): " = BASELINE configs[4]"}.get((a.sampler, a.target, a.dim, a.chains, a.n_sk), "")
    shard = "" if world == 1 else f", sharded over {world} GPUs ({C_local} chains on rank 0)"
    return (f"{a.sampler} d={a.dim} {a.target} target, one ensemble of {a.chains} chains x n_sk={a.n_sk}{base}{shard}; "
            f"grid_size={sampler.grid_size}, tmax={sampler.tmax}, {'adaptive' if sampler.adaptive else 'fixed horizon'}, {slopes}{rr}")


def main_multi(a):
    """`--driver multi`: the sharded step through pdmp_multi_sample_skeleton -- one process, one engine handle and stream per device,
    the final exchange by grouped in-place ncclAllGather inside the library (csrc/multi_device.hip).  Same workload, same JSON line."""
    import ctypes as C
    import torch
    import pdmpflux.jl_amd as P
    from pdmpflux.jl_amd import _lib, ensemble
    n = a.gpus
    assert torch.cuda.is_available() and torch.cuda.device_count() >= n, f"--driver multi --gpus {n}: {torch.cuda.device_count()} device(s) visible"
    sampler = make_sampler(P, a)
    C_total, d, n_sk = a.chains, a.dim, a.n_sk
    gather = 1 if (a.gather == "skeleton" and n > 1) else 0
    L = _lib.lib()
    cfg = sampler._config()
    m = C.c_void_p()
    _lib.check(L.pdmp_multi_create(C.byref(cfg), n, None, C.byref(m)))
    bufs, inits = [], []
    for r in range(n):
        lo, hi = ensemble.shard_range(C_total, r, n)
        rows = C_total if gather else hi - lo
        dev = torch.device("cuda", r)
        bufs.append({"X": torch.empty((rows, n_sk, d), dtype=torch.float64, device=dev), "V": torch.empty((rows, n_sk, d), dtype=torch.float64, device=dev),
                     "t": torch.empty((rows, n_sk), dtype=torch.float64, device=dev)})
        inits.append((torch.empty((max(hi - lo, 1), d), dtype=torch.float64, device=dev), torch.empty((max(hi - lo, 1), d), dtype=torch.float64, device=dev)))
    arr = lambda k: (C.c_void_p * n)(*[b[k].data_ptr() for b in bufs])
    AX, AV, At = arr("X"), arr("V"), arr("t")
    IX, IV = (C.c_void_p * n)(*[q[0].data_ptr() for q in inits]), (C.c_void_p * n)(*[q[1].data_ptr() for q in inits])
    # the shards' initial states generated ON their devices (stream 1 of each chain's key): nothing crosses PCIe per step
    xm, xs, vk = init_kind(a)
    _lib.check(L.pdmp_multi_init_states(m, C_total, 0, C.c_uint64(a.seed), xm, xs, vk, IX, IV))

    def step(i):
        _lib.check(L.pdmp_multi_sample_skeleton_device(m, C_total, n_sk, 0, IX, IV, C.c_uint64(a.seed + i), gather, AX, AV, At))
    for i in range(a.warmup):
        step(i)
    k_ms, x_ms = [], []
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)                                      # returns after every device's stream has drained
        km, xm = C.c_double(), C.c_double()
        _lib.check(L.pdmp_multi_get_timing(m, C.byref(km), C.byref(xm)))
        k_ms.append(km.value); x_ms.append(xm.value)
    elapsed = time.perf_counter() - t0
    tt = bufs[0]["t"][:256]
    assert bool((tt[:, 1:] > tt[:, :-1]).all()) and bool(torch.isfinite(bufs[0]["X"][:256]).all())
    if gather:                                                  # every device holds every shard: compare device n-1's copy with device 0's
        assert torch.equal(bufs[0]["t"].cpu(), bufs[n - 1]["t"].cpu())
    L.pdmp_multi_destroy(m)
    events_step = C_total * (n_sk - 1)
    bytes_per_event = 8 * (2 * d + 1)
    lo0, hi0 = ensemble.shard_range(C_total, 0, n)
    kernel_ms = sum(k_ms) / len(k_ms)                           # the shard kernel, max over devices
    achieved = (hi0 - lo0) * (n_sk - 1) * bytes_per_event / (kernel_ms * 1e-3) / 1e9
    xbytes = (C_total - (hi0 - lo0)) * n_sk * bytes_per_event if gather else 0
    res = {"metric": metric_label(a, sampler, C_total), "value": events_step * a.steps / elapsed, "unit": "events/s", "n_gpus": n, "steps": a.steps,
           "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": workload_label(a, sampler, n, hi0 - lo0), "driver": "multi: one process, pdmp_multi_sample_skeleton (csrc/multi_device.hip)",
                      "chains": C_total, "chains_rank0": hi0 - lo0, "n_sk": n_sk, "dim": d, "events_per_step": events_step,
                      "exchange": "none" if not gather else f"grouped in-place ncclAllGather of X, V, t inside the library: {xbytes} bytes received per GPU per step",
                      "exchange_bytes_per_gpu": xbytes, "kernel_ms_max_over_devices": kernel_ms, "exchange_ms_max_over_devices": sum(x_ms) / len(x_ms),
                      "inits": "device-resident shards from pdmp_multi_init_states (pdmp_multi_sample_skeleton_device: no PCIe staging per step)"},
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                        "kernel_ms": kernel_ms, "bytes_per_event": bytes_per_event,
                        "note": "kernel_ms = launch begin to kernel end on the slowest device (hipEvents on each device's stream)"}}
    print(json.dumps(res), flush=True)


def self_launch(a):
    """`python bench.py --gpus N` (N > 1) without a launcher around it: start the N rank processes ourselves -- a child
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` --
    BEFORE this process has touched the GPU, pass its output through and exit with its code (the line the driver reads is rank 0's)."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()                              # (counting devices does not initialise the GPU)
    if have < a.gpus and os.environ.get("PDMP_BENCH_REHEARSAL") != "1":
        raise SystemExit(f"bench.py --gpus {a.gpus}: {have} GPU(s) visible.  (PDMP_BENCH_REHEARSAL=1 runs the N ranks on device 0 with the exchange "
                         "over gloo -- a rehearsal of the code path on a one-GPU box, not a measurement)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env, cwd=ROOT)


def main():
    a = parse()
    if a.driver == "multi":
        return main_multi(a)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    live = live_counters(a) if (a.pmc == "auto" and int(os.environ.get("WORLD_SIZE", "1")) == 1 and a.gpus == 1) else None
    import torch
    import pdmpflux.jl_amd as P
    from pdmpflux.jl_amd import ensemble

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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
    x, v = make_inputs(ensemble, sampler, a, dev)
    C_total, d, n_sk = a.chains, a.dim, a.n_sk
    lo, hi = ensemble.shard_range(C_total, rank, world)
    C_max = ensemble.shard_range(C_total, 0, world)[1]                 # the largest shard (rank 0)
    # ONE layout for all ranks (the order of the sums over coordinates depends on it): from the largest shard
    if a.lanes == "auto":
        G_auto = sampler.suggest_group_lanes(C_max)
        if G_auto != sampler.group_lanes():
            sampler.set_group_lanes(G_auto)
    elif a.lanes != "default":
        sampler.set_group_lanes(int(a.lanes))
    gather = a.gather if world > 1 else "none"
    n_samples = a.n_samples if a.n_samples > 0 else a.n_sk
    pipeline = 1
    if world > 1 and gather == "skeleton":
        pipeline = 4 if a.pipeline == "auto" else max(1, int(a.pipeline))
    # Output buffers.  N = 1: the timed steps ROTATE over several fresh allocations of X, V, t (as many as fit, at most 3) -- a launch's
    # speed depends on where its arrays lie, so `value` is taken over all of them, none picked -- from the product's own
    # allocator for large outputs (ensemble.alloc_outputs: independent 1-GiB physical chunks).  N > 1: ONE full-size set per rank from
    # torch's allocator: the shard is sampled straight into its chain range and gathered in place (the buffers go through RCCL).
    rows = C_total if gather == "skeleton" else hi - lo        # (gather none / samples / moments: shard-sized skeleton buffers)
    set_bytes = 8 * rows * n_sk * (2 * d + 1)
    allocator = a.alloc if a.alloc != "auto" else ("chunked" if world == 1 else "hipmalloc")

    alloc_log = []                                               # per allocation: what the product's allocator drew and measured

    def alloc_set(kind):
        if kind in ("chunked", "chunked_one_draw"):              # the product's allocator (up to three draws, the fastest kept) / a single draw
            o = ensemble.alloc_outputs(rows, n_sk, d, device=dev, chunked=True, draws=None if kind == "chunked" else 1)
            alloc_log.append(dict(ensemble.last_alloc_info))
            return o
        alloc_log.append({})
        return {"X": torch.empty((rows, n_sk, d), dtype=torch.float64, device=dev), "V": torch.empty((rows, n_sk, d), dtype=torch.float64, device=dev),
                "t": torch.empty((rows, n_sk), dtype=torch.float64, device=dev)}

    def run_step(i, full):
        if world == 1:
            ensemble.sample_skeleton_device(sampler, n_sk, x, v, seed=a.seed + i, out=full, check=False)
        else:       # rank r samples chains [lo_r, hi_r); gather="skeleton": then one all-gather each for X, V, t (pipeline > 1: per batch of chains,
                    # overlapping the next batch's kernel); "samples" / "moments": sample_from_skeleton on the shard, then the exchange of its result
            return ensemble.sample_skeleton_sharded(sampler, n_sk, x, v, seed=a.seed + i, gather=gather, out_full=full, check=False,
                                                    via_host=rehearsal, pipeline=pipeline, N_samples=n_samples if gather in ("samples", "moments") else None)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_launches(full, n, seed0):                          # hipEvent-timed launches on torch's current stream (the engine's launch stream)
        evs = []
        for i in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run_step(seed0 + i, full); e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        return [e0.elapsed_time(e1) for e0, e1 in evs]

    free_b = torch.cuda.mem_get_info()[0]
    n_sets = max(1, min(a.sets if a.sets > 0 else 1, int(0.9 * free_b // max(set_bytes, 1))))
    sets = [alloc_set(allocator) for _ in range(n_sets)]         # the FIRST allocation(s) of this process, whatever they turn out to be
    full = sets[0]

    last_step = []

    def step(i):
        r = run_step(i, sets[i % n_sets])
        last_step[:] = [r]

    # The pipelined exchange has run on one GPU only (RCCL with one rank, gloo rehearsals): if its first step fails on any rank, every rank
    # falls back to the plain exchange after the whole shard -- decided together, so that no rank waits in a collective the others left.
    if world > 1 and pipeline > 1:
        ok = 1
        try:
            step(0)
            torch.cuda.synchronize()
        except Exception as e:                                    # noqa: BLE001
            ok = 0
            print(f"[bench rank {rank}] pipelined exchange failed ({type(e).__name__}: {e}); falling back to --pipeline 1", file=sys.stderr, flush=True)
        flag = torch.tensor([ok], dtype=torch.int32, device="cpu" if rehearsal else dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            pipeline = 1
    for i in range(a.warmup):
        step(i)
    fence()
    ensemble.kernel_timing(sampler, reset=True)
    step_ev = []
    t0 = time.perf_counter()
    for i in range(a.steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        step(a.warmup + i)
        e1.record()
        step_ev.append((e0, e1))
    fence()
    elapsed = time.perf_counter() - t0
    step_ms = [e0.elapsed_time(e1) for e0, e1 in step_ev]        # per step, on the launch stream: which allocation set was slow, if any
    placement = []
    for k in range(n_sets):
        mine = [m for i, m in enumerate(step_ms) if (a.warmup + i) % n_sets == k]
        if mine:
            placement.append({"allocator": allocator, "in_timed_region": True, "launches": len(mine), "kernel_ms": [round(m, 3) for m in mine], "alloc": alloc_log[k]})
    from pdmpflux.jl_amd import _lib
    _lib.check(_lib.lib().pdmp_check_status(sampler.handle(), None))
    tm = ensemble.kernel_timing(sampler)
    kernel_ms = tm["kernel_ms_total"] / max(a.steps, 1)        # kernel time per step on this rank: every launch of the step (a pipelined step makes one per batch;
                                                                # --gather samples / moments add their sample_from_skeleton / moments kernel)
    launches_per_step = tm["launches"] / max(a.steps, 1)
    ab = None
    if a.ab_env and world == 1:
        name, vals = a.ab_env.split("=", 1)
        vals = vals.split(",")
        before = os.environ.get(name)
        ab = {"knob": name, "ms": {val: [] for val in vals}}
        for rnd in range(3):
            for val in vals:
                os.environ[name] = val
                sampler.reload_policy()                       # (the PDMP_* knobs are part of the handle's launch policy: no launch reads the environment)
                ab["ms"][val] += [round(m, 3) for m in timed_launches(full, 2, 7000 + 10 * rnd)]
        if before is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = before
        sampler.reload_policy()
        ab["mean_ms"] = {val: round(sum(m) / len(m), 3) for val, m in ab["ms"].items()}
    kernel_ms_max = kernel_ms
    if world > 1:
        te = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms_max = float(te[0].item()), float(te[1].item())
    n_batches = min(pipeline, max(1, C_total // world)) if pipeline > 1 else 1

    def ranges_of(r):                                             # the chains rank r samples
        if n_batches > 1:
            return [ensemble.batch_shard_range(C_total, j, n_batches, r, world) for j in range(n_batches)]
        return [ensemble.shard_range(C_total, r, world)]
    C_local = sum(h2 - l2 for l2, h2 in ranges_of(rank))

    # sanity inside the bench: strictly increasing event times on a slice, finite outputs; after the exchange every rank holds
    # every rank's chains: the blocks of every rank must be what a checksum by that rank says
    l0, h0 = ranges_of(rank)[0]
    own = full["t"][l0:h0] if gather == "skeleton" else full["t"]
    tt = own[:256]
    if not (os.environ.get("PDMP_LANE_DEBUG", "0") != "0" or os.environ.get("PDMP_DENSE_DEBUG", "0") != "0"):     # (measurement knobs that drop stores: nothing to check)
        assert bool((tt[:, 1:] > tt[:, :-1]).all()) and bool(torch.isfinite((full["X"][l0:h0] if gather == "skeleton" else full["X"])[:256]).all())
    exchange_ms = None
    if gather == "skeleton":
        cdev = "cpu" if rehearsal else dev

        def checksum(r):
            parts = [torch.stack([full["t"][l2:h2].sum(), full["X"][l2:h2, -1].sum(), full["V"][l2:h2, -1].sum()]) for l2, h2 in ranges_of(r) if h2 > l2]
            return torch.stack(parts).sum(0)
        mine = checksum(rank).to(cdev)
        sums = [torch.zeros(3, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(sums, mine)
        for r in range(world):
            assert torch.equal(checksum(r).cpu(), sums[r].cpu()), f"rank {rank}: the gathered blocks of rank {r} differ from what that rank computed"
        # the exchange ALONE (untimed calibration, after the timed region): the three all-gathers of one step on the buffers as they stand
        fence()
        tx = time.perf_counter()
        for k in ("X", "V", "t"):
            ensemble._gather_in_place(full[k], C_total, rank, world, None, via_host=rehearsal)
        fence()
        exchange_ms = (time.perf_counter() - tx) * 1e3
    elif world > 1 and gather in ("samples", "moments") and last_step:
        res_last = last_step[-1]
        if gather == "moments":
            assert bool(torch.isfinite(res_last["mean"]).all()) and float(res_last["count"]) == float(C_total * n_samples)
        else:
            assert tuple(res_last["samples"].shape) == (C_total, n_samples, d) and bool(torch.isfinite(res_last["samples"][:64]).all())

    # placement survey (N = 1; untimed, after everything `value` depends on): the timed arrays are released and five more fresh
    # allocations of X, V, t are timed one after the other (one warm + three timed launches each)
    if a.placement_survey == "auto" and world == 1 and a.steps >= 5 and set_bytes >= (1 << 30):
        del full, own, tt
        sets.clear()
        torch.cuda.empty_cache()
        ensemble.pool_trim()                                      # (the survey wants FRESH allocations: nothing may come back from the allocator's pool)
        for kind in ("chunked", "hipmalloc", "chunked_one_draw", "hipmalloc", "chunked"):
            cur = alloc_set(kind)
            ms = timed_launches(cur, 4, 1000 + 10 * len(placement))[1:]
            placement.append({"allocator": kind, "in_timed_region": False, "launches": len(ms), "kernel_ms": [round(m, 3) for m in ms], "alloc": alloc_log[-1]})
            del cur
            torch.cuda.empty_cache()
            ensemble.pool_trim()
    if rank == 0:
        events_step = C_total * (n_sk - 1)
        value = events_step * a.steps / elapsed
        bytes_per_event = 8 * (2 * d + 1)                         # X, V, t of one event
        achieved = C_local * (n_sk - 1) * bytes_per_event / (kernel_ms * 1e-3) / 1e9
        G = sampler.group_lanes()
        K = (d + G - 1) // G if G > 0 else 0
        # mirrors RegForm in csrc/group_kernel.hpp (which register-form instantiations exist)
        sid, fd = sampler.sampler_id, sampler.deriv_mode != 0
        small_k = (G >= 2 and K < 7)
        reg_form = (G > 0 and sid in (0, 1, 2) and K <= 13 and G <= 16 and K >= 2 and a.target in ("iso", "diag", "banana")
                    and not (sid == 0 and fd) and not (sid == 2 and K != 13 and K < 7) and not (small_k and (fd or sid == 2))
                    and os.environ.get("PDMP_REG", "1") != "0")
        if sid == 5:                                              # RHMC (round 4): FULL 13-slot kernels on iso / banana, partial ones on diag
            reg_form = (G > 0 and G <= 16 and os.environ.get("PDMP_REG", "1") != "0" and
                        ((a.target in ("iso", "banana") and K == 13) or (a.target == "diag" and K <= 13 and (G == 1 or K >= 7))))
        if sid == 3:                                              # ForwardECMC (round 4): exact-K kernels, K <= 7 (five LDS tiles per wave beside the registers)
            reg_form = (G > 0 and G <= 16 and not fd and a.target in ("iso", "diag", "banana") and K <= 7 and K >= (2 if G == 1 else 4)
                        and os.environ.get("PDMP_REG", "1") != "0" and os.environ.get("PDMP_REG_EXACT", "1") != "0")
        kernel = (f"group_kernel_reg<{sampler.name},{a.target}> (G={G} lanes per chain, K={K} slots, x and v in registers)" if reg_form else
                  f"group_kernel<{sampler.name},{a.target}> (G={G} lanes per chain)" if G > 0 else
                  f"skeleton_kernel<{sampler.name},{a.target}> (wave per chain)")
        form = sampler.last_launch_form()                        # which kernel form served the timed launches (pdmp_last_launch_form)
        if form == 3:                                            # (the engine's own report wins over the mirror above)
            kernel = f"group_kernel_reg<{sampler.name},{a.target}> (G={G} lanes per chain, K={K} slots, x and v in registers)"
        elif form == 2:
            kernel = f"group_kernel<{sampler.name},{a.target}> (G={G} lanes per chain)"
        if form in (4, 7):
            per = "one lane per chain, 64 chains per wave" if form == 4 else "a lane pair per chain, 32 chains per wave"
            kernel = (f"lane_kernel<{a.target}> ({per}; signed positions y = x v in registers, "
                      "events stored as whole 64-byte lines)")
        cnt = measured_counters(a, C_local, G, form)
        if live:                                                  # measured now, on this GPU, for this workload
            events_launch = C_local * (n_sk - 1)
            cnt = {"hbm_bytes_per_launch": live["WRITE_SIZE"] * 1024 + live["FETCH_SIZE"] * 1024 * 2,
                   "write_bytes": live["WRITE_SIZE"] * 1024, "read_bytes_corrected": live["FETCH_SIZE"] * 1024 * 2,
                   "source": "LIVE: rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE (KiB; FETCH_SIZE x2 on gfx950) around one launch of this workload in child "
                             "processes of this run, each counter in its own pass; kernels " + ", ".join(live.get("kernels", []))}
            if live.get("SQ_INSTS_VALU") and live.get("GRBM_GUI_ACTIVE"):
                cnt.update({"valu_instr_per_event": live["SQ_INSTS_VALU"] / events_launch,
                            "salu_instr_per_event": live.get("SQ_INSTS_SALU", 0.0) / events_launch,
                            "valu_busy": 4 * live["SQ_ACTIVE_INST_VALU"] / (live["GRBM_GUI_ACTIVE"] / 8 * 1024),
                            "valu_source": "LIVE: SQ_INSTS_VALU / events of one launch; busy = 4 SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)"})
        xbytes = ((C_total - C_local) * n_sk * bytes_per_event if gather == "skeleton" else
                  (C_total - C_local) * n_samples * d * 8 if gather == "samples" else
                  (2 * d + 1) * 8 if gather == "moments" else 0)
        xdesc = {"none": "none",
                 "skeleton": f"one all-gather each of the X, V, t shards per step (RCCL, in place{', in %d batches of chains overlapping the sampling' % n_batches if n_batches > 1 else ''}): {xbytes} bytes received per GPU per step",
                 "samples": f"sample_from_skeleton({n_samples}) on every shard, one all-gather of the (chains, {n_samples}, dim) samples: {xbytes} bytes received per GPU per step",
                 "moments": f"fused moment sums of sample_from_skeleton({n_samples}) on every shard (pdmp_sample_moments_device), one all-reduce of {2 * d + 1} doubles"}[gather]
        res = {
            "metric": metric_label(a, sampler, C_total),
            "value": value, "unit": "events/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload_label(a, sampler, world, C_local), "chains": C_total, "chains_rank0": C_local,
                       "n_sk": n_sk, "dim": d, "events_per_step": events_step, "lanes_per_chain": G,
                       "exchange": xdesc, "gather": gather, "pipeline_batches": n_batches,
                       "exchange_bytes_per_gpu": xbytes,
                       "exchange_ms": exchange_ms,
                       "exchange_note": (None if exchange_ms is None else
                                         "exchange_ms: the three all-gathers of one step ALONE, timed once after the timed region (wall clock between two barriers); "
                                         "ms_per_step covers sampling and exchange together (overlapped when pipeline_batches > 1); "
                                         "kernel_only_events_per_s = events per step / the slowest rank's kernel time per step"),
                       "kernel_launches_per_step": launches_per_step,
                       "allocator": ("ensemble.alloc_outputs: X, V, t built from independent 1-GiB physical chunks, the set drawn up to three times and the one a store-pattern kernel writes fastest kept (pdmp_device_alloc_streams)" if allocator == "chunked"
                                     else "torch.empty (hipMalloc)"),
                       "allocation_sets": n_sets,
                       "placement_ms": placement,
                       "placement_note": ("per fresh allocation of X, V, t: hipEvent-timed launches (ms, incl. the exchange when N > 1).  `value` covers every step of the "
                                          "timed region = the FIRST allocation(s) this process made (in_timed_region true; a rule, not a choice); the others are the untimed "
                                          "survey run afterwards: five more fresh allocations -- two from the product's allocator, one single draw of it without selection (chunked_one_draw), two from hipMalloc; `alloc`: the draws the allocator timed with its store-pattern kernel, ms, and what the allocation cost in seconds (the placement "
                                          "lottery of DESIGN.md 6).  all_allocations_ms: mean launch time over every allocation listed"),
                       "all_allocations_ms": (sum(sum(q["kernel_ms"]) / len(q["kernel_ms"]) for q in placement) / len(placement)) if placement else None,
                       "kernel_ms_max_over_ranks": kernel_ms_max,
                       "kernel_only_events_per_s": events_step / (kernel_ms_max * 1e-3)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": cnt.get("hbm_bytes_per_launch") if cnt else None,
                         "traffic_source": (cnt.get("source", "profiles/traffic.json") + ("" if live else " -- committed rocprofv3 measurement of this workload, not read in this run")) if cnt else None,
                         "traffic_write": cnt.get("write_bytes") if cnt else None, "traffic_read": cnt.get("read_bytes_corrected") if cnt else None,
                         "kernel": kernel, "kernel_ms": kernel_ms, "bytes_per_event": bytes_per_event,
                         "valu": ({"instr_per_event": cnt.get("valu_instr_per_event"), "salu_instr_per_event": cnt.get("salu_instr_per_event"), "busy": cnt.get("valu_busy"),
                                   "source": cnt.get("valu_source")} if cnt and cnt.get("valu_instr_per_event") else None),
                         "note": ("bound by the instruction stream of the one wave per SIMD: a wave alone issues an fp64 vector instruction every ~8 cycles, and "
                                  "roofline.valu.instr_per_event x 64 chains x 8 cycles is the launch time to the percent (DESIGN.md 4.1b) -- `busy`, computed at 4 cycles "
                                  "per instruction, under-states the SIMD's utilisation by that factor; since round 5 the kernel writes whole 128-byte lines and "
                                  "roofline.traffic is 1.02 x the events' bytes (a plain fill of this GPU reaches ~5.6 - 6.3 TB/s)"
                                  if form in (4, 7) else
                                  "the kernel is bound by fp64 VALU issue, not by HBM (roofline.valu: wave-instructions per event and the share of "
                                  "SIMD cycles the VALUs are busy, from the committed PMC pass); the HBM fraction is the requested headline")},
        }
        # The ceiling that binds the VALU-bound workloads (BPS, Boomerang, small shards): fp64 vector issue.  Every VALU wave-instruction
        # occupies its SIMD's 16 fp64 lanes for 4 cycles whatever it computes, so SQ_INSTS_VALU x 64 lanes x 2 flop (an FMA per lane and
        # issue slot) is the flop-equivalent of the issue slots the launch used; against the 78.6 TFLOP/s fp64 vector peak
        # (256 CUs x 4 SIMDs x 16 lanes x 2 x 2.4 GHz) that is the share of all VALU issue slots of the chip at its top clock.
        if cnt and cnt.get("valu_instr_per_event"):
            slots = cnt["valu_instr_per_event"] * C_local * (n_sk - 1) * 64 * 2 / (kernel_ms * 1e-3) / 1e12
            res["roofline_valu"] = {"bound": "valu-f64", "achieved": slots, "peak": 78.6, "unit": "TFLOP/s", "frac": slots / 78.6,
                                    "instr_per_event": cnt["valu_instr_per_event"], "busy": cnt.get("valu_busy"),
                                    "note": "achieved = SQ_INSTS_VALU x 64 lanes x 2 flop / kernel time: the fp64-FMA equivalent of the VALU issue slots used "
                                            "(not all of them hold fp64 arithmetic); peak = fp64 vector peak at 2.4 GHz; `busy` is the same share against the "
                                            "clock the chip actually held (4 SQ_ACTIVE_INST_VALU / SIMD cycles)"}
        if a.target == "dense" and int(_lib.lib().pdmp_dense_batched_last_mode()) == 2:
            gemm = "rocBLAS dgemm" if os.environ.get("PDMP_DENSE_GEMM", "mfma") == "rocblas" else "fp64 MFMA product (csrc/dgemm_mfma.hip)"
            res["roofline"]["kernel"] = ("di_segment_kernel (wave per chain; P x and P v kept current by O(d) updates, one " + gemm + " over all "
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
        # beside `value`, at the top level: which allocator the timed arrays came from and what it cost (the selection runs before the timed region)
        if ab:
            res["ab_env"] = ab
        res["allocator"] = allocator
        res["alloc_seconds"] = [q.get("seconds") for q in alloc_log[:n_sets]]
        res["all_allocations_ms"] = res["config"]["all_allocations_ms"]
        if a.cpu_seconds > 0 and world == 1:                     # CPU baseline: rank 0 at N = 1 only
            res["cpu_baseline"] = cpu_baseline(a, sampler)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
